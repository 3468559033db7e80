#!/usr/bin/env python3
"""Differential fuzz of the TEST-ONLY oracle (oracle/mp2_oracle.c) against the REAL reference (oracle/_ref/libtoolame_ref.so, the
reference's own sources compiled where they lie): random legal (rate, mode, bitrate) x psy 0..4 x every signal kind of
tests/pcmgen.py x random seeds, plus the degenerate sweep -- a lone impulse and full-scale square waves of every period 2..64 under
psy 1 / 2 / 4 in six configurations -- where hundreds of spectral lines of nearly equal level make one ulp decide a tone test or an
allocation tie.  Compared: every byte and every per-call return length (the 4096-byte burst cadence), toolame_finish() included.

  tools/fuzz_oracle_vs_ref.py [--streams N] [--frames F] [--seed S] [--sweep] [--jobs J]      (build container only: needs oracle/_ref)

The reference is a process-global singleton (toolame.c:24-26,89-118), so each stream gets a FRESH copy of it: a worker dlopen()s the
library, encodes one stream, dlclose()s it (its statics go with it).  Inputs on which the reference itself is undefined are left out:
psy 3 on digital silence / a lone impulse divides 0 by 0 for an array index (psycho_3.c:299) and crashes.
Exit status 1 on any mismatch.  Record of the long run: profiles/fuzz_oracle_vs_ref_r04.txt."""
import argparse
import ctypes as C
import os
import sys
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import oraclelib as O  # noqa: E402
from pcmgen import gen_pcm  # noqa: E402

# legal Layer II bitrates per MPEG version and what the mode allows (ISO 11172-3 2.4.2.3: 32..56 mono only, 224..384 not mono)
V1 = (32, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384)
V2 = (8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160)
RATES = (48000, 44100, 32000, 24000, 22050, 16000)


def legal(fs, mode, kbps):
    if fs >= 32000:
        if kbps not in V1: return False
        if mode == "m": return kbps <= 192
        return kbps not in (32, 48, 56, 80)
    return kbps in V2


def one_reference(job):
    """-> (data, lens) of the real reference for one stream; a fresh library instance per stream"""
    import _ctypes
    fs, mode, kbps, psy, kind, seed, F = job
    pcm = gen_pcm(seed, kind, 0, F)
    L = C.CDLL(str(O.REF_SO))
    try:
        L.toolame_set_samplerate.argtypes = [C.c_long]; L.toolame_set_channel_mode.argtypes = [C.c_char]
        L.toolame_encode_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.toolame_finish.argtypes = [C.c_void_p, C.c_size_t]
        rc = [L.toolame_init(), L.toolame_set_samplerate(fs), L.toolame_set_psy_model(min(psy, 3)), L.toolame_set_channel_mode(mode.encode()),
              L.toolame_set_bitrate(kbps), L.toolame_set_pad(0)]
        assert rc == [0] * 6, (job, rc)
        if psy > 3:
            C.c_int.in_dll(L, "tlref_model").value = psy          # the setter refuses model 4 (toolame.c:204-207)
        out = (C.c_ubyte * 4096)()
        chunks, lens = [], []
        for f in range(F):
            buf = np.ascontiguousarray(pcm[f])
            n = L.toolame_encode_frame(buf.ctypes.data, None, 0, out, 4096)
            chunks.append(bytes(out[:n])); lens.append(n)
        n = L.toolame_finish(out, 4096)
        chunks.append(bytes(out[:n])); lens.append(n)
    finally:
        h = L._handle
        del L
        _ctypes.dlclose(h)
    return b"".join(chunks), lens


def compare(job):
    fs, mode, kbps, psy, kind, seed, F = job
    ref = one_reference(job)
    got = O.oracle_stream(gen_pcm(seed, kind, 0, F), samplerate=fs, mode=mode, kbps=kbps, psy=psy)
    return job, ref[0] == got[0] and list(ref[1]) == list(got[1])


def _quiet():
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 2)                                           # the reference prints its table choice to stderr on every init


def random_jobs(n, F, seed):
    rng = np.random.default_rng(seed)
    jobs = []
    while len(jobs) < n:
        fs = int(rng.choice(RATES)); mode = "sjdm"[rng.integers(4)]
        kbps = int(rng.choice(V1 if fs >= 32000 else V2))
        if not legal(fs, mode, kbps): continue
        psy = int(rng.integers(5)); kind = int(rng.integers(8))
        if psy == 3 and kind in (1, 3): continue                  # the reference crashes here (psycho_3.c:299)
        jobs.append((fs, mode, kbps, psy, kind, int(rng.integers(1 << 30)), F))
    return jobs


def sweep_jobs(F):
    cfgs = [(48000, "s", 128), (48000, "j", 192), (48000, "m", 64), (32000, "s", 128), (24000, "m", 64), (44100, "j", 160)]
    jobs = []
    for fs, mode, kbps in cfgs:
        for psy in (1, 2, 4):
            jobs.append((fs, mode, kbps, psy, 3, 1, F))            # lone impulse
            for period in range(2, 65):                            # kind 2: full-scale square wave of period 2 + seed % 63 ... both phases
                jobs.append((fs, mode, kbps, psy, 2, period - 2, F))
    return jobs


def run(jobs, njobs):
    bad = []
    with ProcessPoolExecutor(max_workers=njobs, initializer=_quiet) as ex:
        for job, ok in ex.map(compare, jobs, chunksize=16):
            if not ok: bad.append(job)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=2000)
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--sweep", action="store_true", help="add the impulse / square-wave sweep (1152 streams)")
    ap.add_argument("--jobs", type=int, default=os.cpu_count() or 4)
    a = ap.parse_args()
    if not O.REF_SO.exists():
        sys.exit("oracle/_ref/libtoolame_ref.so is not built here (make -C oracle ref: needs /root/reference)")
    O.build_oracle()
    jobs = random_jobs(a.streams, a.frames, a.seed) + (sweep_jobs(a.frames) if a.sweep else [])
    bad = run(jobs, a.jobs)
    by_psy = {p: sum(1 for j in jobs if j[3] == p) for p in range(5)}
    print(f"oracle vs live reference: {len(jobs)} streams x {a.frames} frames (random {a.streams}, seed {a.seed}" + (", + sweep 1152" if a.sweep else "") +
          f"), per psy model {by_psy}: {len(bad)} mismatching streams", bad[:10])
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
