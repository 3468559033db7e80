#!/usr/bin/env python3
"""Differential fuzz of the TEST-ONLY oracle (oracle/mp2_oracle.c) against the REAL reference (oracle/_ref/libtoolame_ref.so, the
reference's own sources compiled where they lie): random legal (rate, mode, bitrate) x psy 0..4 x every signal kind of
tests/pcmgen.py x random seeds, plus the degenerate sweep -- a lone impulse and full-scale square waves of every period 2..64 under
psy 1 / 2 / 4 in six configurations -- where hundreds of spectral lines of nearly equal level make one ulp decide a tone test or an
allocation tie.  Compared: every byte and every per-call return length (the 4096-byte burst cadence), toolame_finish() included.
Round 6: X-PAD is fuzzed too -- `--xpad P` gives that share of the random streams a PAD length (toolame_set_pad: 2..255, every length the
caller accepts, src/odr-audioenc.cpp:566, that leaves room for the frame's header, CRC and bit allocation -- the product's own rule,
csrc/mp2_host.cpp) and every frame a random X-PAD length in {0, 2..pad_len} with random bytes (toolame.c:301,515-551: the bit budget,
the X-PAD bytes, the F-PAD taken from the caller's buffer).

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
    fs, mode, kbps, psy, kind, seed, F, pad_len = job
    pcm = gen_pcm(seed, kind, 0, F)
    xp = xpads_for(seed, pad_len, F)
    L = C.CDLL(str(O.REF_SO))
    try:
        L.toolame_set_samplerate.argtypes = [C.c_long]; L.toolame_set_channel_mode.argtypes = [C.c_char]
        L.toolame_encode_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.toolame_finish.argtypes = [C.c_void_p, C.c_size_t]
        rc = [L.toolame_init(), L.toolame_set_samplerate(fs), L.toolame_set_psy_model(min(psy, 3)), L.toolame_set_channel_mode(mode.encode()),
              L.toolame_set_bitrate(kbps), L.toolame_set_pad(pad_len)]
        assert rc == [0] * 6, (job, rc)
        if psy > 3:
            C.c_int.in_dll(L, "tlref_model").value = psy          # the setter refuses model 4 (toolame.c:204-207)
        out = (C.c_ubyte * 4096)()
        chunks, lens = [], []
        for f in range(F):
            buf = np.ascontiguousarray(pcm[f])
            if xp is None:
                n = L.toolame_encode_frame(buf.ctypes.data, None, 0, out, 4096)
            else:
                xb = (C.c_ubyte * len(xp[f][0])).from_buffer_copy(xp[f][0])
                n = L.toolame_encode_frame(buf.ctypes.data, xb, xp[f][1], out, 4096)
            chunks.append(bytes(out[:n])); lens.append(n)
        n = L.toolame_finish(out, 4096)
        chunks.append(bytes(out[:n])); lens.append(n)
    finally:
        h = L._handle
        del L
        _ctypes.dlclose(h)
    return b"".join(chunks), lens


def xpads_for(seed, pad_len, F):
    """per frame (pad_len + 1 random bytes, X-PAD length): 0 (no PAD this frame), the full length, or anything from 2 up"""
    if not pad_len:
        return None
    rng = np.random.default_rng(seed ^ 0x5EED)
    out = []
    for _ in range(F):
        u = rng.random()
        xl = 0 if u < 0.15 else pad_len if u < 0.5 else int(rng.integers(2, pad_len + 1))
        out.append((bytes(rng.integers(0, 256, pad_len + 1, dtype=np.uint8)), xl))
    return out


def oracle_one(job):
    fs, mode, kbps, psy, kind, seed, F, pad_len = job
    pcm = gen_pcm(seed, kind, 0, F)
    xp = xpads_for(seed, pad_len, F)
    if xp is None:
        return O.oracle_stream(pcm, samplerate=fs, mode=mode, kbps=kbps, psy=psy)
    e = O.OracleEncoder(samplerate=fs, mode=mode, kbps=kbps, psy=psy, pad_len=pad_len)
    chunks = [e.encode(pcm[f], xp[f][0], xp[f][1]) for f in range(F)] + [e.finish()]
    e.close()
    return b"".join(chunks), [len(c) for c in chunks]


def compare(job):
    ref = one_reference(job)
    got = oracle_one(job)
    return job, ref[0] == got[0] and list(ref[1]) == list(got[1])


def pad_fits(fs, mode, kbps, pad_len):
    """the product's rule (csrc/mp2_host.cpp tl_build_config): the PAD leaves room for header, CRC-16, bit allocation and ScF-CRC"""
    import emulib as E
    try:
        E.EmuBatch([dict(samplerate=fs, mode=mode, kbps=kbps, psy=1, pad_len=pad_len)]).close()
        return True
    except ValueError:
        return False


def _quiet():
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 2)                                           # the reference prints its table choice to stderr on every init


def random_jobs(n, F, seed, xpad_share=0.0):
    rng = np.random.default_rng(seed)
    jobs = []
    while len(jobs) < n:
        fs = int(rng.choice(RATES)); mode = "sjdm"[rng.integers(4)]
        kbps = int(rng.choice(V1 if fs >= 32000 else V2))
        if not legal(fs, mode, kbps): continue
        psy = int(rng.integers(5)); kind = int(rng.integers(8))
        if psy == 3 and kind in (1, 3): continue                  # the reference crashes here (psycho_3.c:299)
        pad_len = 0
        if xpad_share and rng.random() < xpad_share:
            pad_len = int(rng.choice([2, 4, 58, 196, 255, int(rng.integers(2, 256)), int(rng.integers(2, 64))]))
            while pad_len > 2 and not pad_fits(fs, mode, kbps, pad_len):
                pad_len = max(2, pad_len // 2)                    # small frames: the largest length of this walk that still fits
            if not pad_fits(fs, mode, kbps, pad_len): pad_len = 0
        jobs.append((fs, mode, kbps, psy, kind, int(rng.integers(1 << 30)), F, pad_len))
    return jobs


def sweep_jobs(F):
    cfgs = [(48000, "s", 128), (48000, "j", 192), (48000, "m", 64), (32000, "s", 128), (24000, "m", 64), (44100, "j", 160)]
    jobs = []
    for fs, mode, kbps in cfgs:
        for psy in (1, 2, 4):
            jobs.append((fs, mode, kbps, psy, 3, 1, F, 0))         # lone impulse
            for period in range(2, 65):                            # kind 2: full-scale square wave of period 2 + seed % 63 ... both phases
                jobs.append((fs, mode, kbps, psy, 2, period - 2, F, 0))
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
    ap.add_argument("--xpad", type=float, default=0.0, help="share of the random streams that carry PAD (random length per stream, random X-PAD length per frame)")
    ap.add_argument("--jobs", type=int, default=os.cpu_count() or 4)
    a = ap.parse_args()
    if not O.REF_SO.exists():
        sys.exit("oracle/_ref/libtoolame_ref.so is not built here (make -C oracle ref: needs /root/reference)")
    O.build_oracle()
    jobs = random_jobs(a.streams, a.frames, a.seed, a.xpad) + (sweep_jobs(a.frames) if a.sweep else [])
    bad = run(jobs, a.jobs)
    by_psy = {p: sum(1 for j in jobs if j[3] == p) for p in range(5)}
    npad = sum(1 for j in jobs if j[7])
    print(f"oracle vs live reference: {len(jobs)} streams x {a.frames} frames (random {a.streams}, seed {a.seed}" + (", + sweep 1152" if a.sweep else "") +
          f"), per psy model {by_psy}, {npad} streams with PAD (lengths {sorted({j[7] for j in jobs if j[7]})[:8]} ...): {len(bad)} mismatching streams", bad[:10])
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
