#!/usr/bin/env python3
"""GPU soak (run on the GPU box): N streams with random legal configurations and mixed signals (integer-generated kinds and the
tone-labelling stress signals of tools/fuzz_emu.py), F frames each in ragged chunks, device output against the TEST-ONLY oracle
byte for byte.  usage: tools/soak_gpu.py [nstreams] [nframes] [seed]"""
import sys
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
import odr_audioenc_amd as M
import oraclelib as O
from fuzz_emu import crafted
from pcmgen import gen_pcm

RATES = {48000: [(m, k) for m in "sjdm" for k in ((64, 96, 128, 160, 192, 256, 384) if m != "m" else (32, 48, 64, 96, 128, 192))],
         32000: [("s", 128), ("j", 192), ("m", 64), ("m", 96), ("d", 256)],
         24000: [("s", 64), ("j", 96), ("m", 32), ("m", 64), ("s", 128)],
         16000: [("m", 24), ("s", 48), ("j", 64)],
         44100: [("s", 128), ("j", 192), ("m", 64), ("d", 256), ("s", 384), ("j", 96)],
         22050: [("m", 32), ("s", 64), ("j", 160), ("m", 8), ("s", 128)]}


def ref_of(job):
    pcm, fs, mode, kbps, psy = job
    try:
        return O.oracle_stream(pcm, samplerate=fs, mode=mode, kbps=kbps, psy=psy)[0]
    except Exception:  # noqa: BLE001
        return None


MODELS = [0, 1, 1, 1, 2, 3, 3, 4]      # TL_SOAK_MODELS=0,1,3 restricts the soak to the models with no known last-ulp ties


def main():
    global MODELS
    import os
    if os.environ.get("TL_SOAK_MODELS"):
        MODELS = [int(x) for x in os.environ["TL_SOAK_MODELS"].split(",")]
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rng = np.random.default_rng(seed)
    combos = [(fs, m, k) for fs, lst in RATES.items() for m, k in lst]
    jobs = []
    while len(jobs) < S:
        fs, mode, kbps = combos[rng.integers(len(combos))] if rng.random() < 0.5 else (48000, "sj"[rng.integers(2)], (128, 192)[rng.integers(2)])
        psy = int(rng.choice(MODELS))
        s = int(rng.integers(1 << 30))
        if os.environ.get("TL_SOAK_EDGE"):                # the edge spectra of tools/fuzz_psy2_edge.py (6 frames each, repeated to F)
            import fuzz_psy2_edge as Fz
            e = Fz.make(Fz.KINDS[int(rng.integers(len(Fz.KINDS)))], s)
            pcm = np.concatenate([e] * ((F + 5) // 6))[:F]
        elif rng.random() < 0.4:
            pcm = crafted(s)[:F] if F <= 6 else np.concatenate([crafted(s + i) for i in range((F + 5) // 6)])[:F]
        else:
            kind = int(rng.integers(8))                    # incl. psy 3 on silence / impulse: the reference crashes there, the oracle defines it
            pcm = gen_pcm(s, kind, 0, F)
        jobs.append((np.ascontiguousarray(pcm), fs, mode, kbps, psy))
    O.lib()
    with ProcessPoolExecutor() as ex:
        refs = list(ex.map(ref_of, jobs, chunksize=8))
    keep = [i for i, r in enumerate(refs) if r is not None]
    jobs, refs = [jobs[i] for i in keep], [refs[i] for i in keep]
    b = M.Batch([M.StreamConfig(samplerate=j[1], mode=j[2], bitrate=j[3], psy_model=j[4]) for j in jobs])
    pcm = np.stack([j[0] for j in jobs], axis=1)
    chunks, pos = [b""] * len(jobs), 0
    for n in (1, 3, F - 4):
        if n <= 0:
            continue
        got, _ = b.encode(pcm[pos:pos + n])
        chunks = [a + c for a, c in zip(chunks, got)]
        pos += n
    tail = b.flush()
    bad = [(i, jobs[i][1:]) for i in range(len(jobs)) if chunks[i] + tail[i] != refs[i]]
    print(f"{len(jobs)} streams x {pos} frames (seed {seed}): {len(bad)} mismatching streams", bad[:10])
    b.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
