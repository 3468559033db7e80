#!/usr/bin/env python3
"""Edge-signal fuzz of psy models 2 and 4 on the HOST EMULATION of the kernel source against the oracle (CPU only): spectra that
drive the restated glibc routines (csrc/tl_libm.h) into their special branches -- arctangent with a zero operand, extreme operand
ratios, both evaluation forms; sincos at 0 / pi / multiples of pi/4; energies at and below the 0.0005 clamp -- DC, Nyquist, single
impulses at every kind of position, exact-bin sinusoids (period 2^k samples), one-LSB signals, two-sample patterns, sparse frames.
usage: tools/fuzz_psy2_edge.py [cases per kind, default 40] [seed] [models, default 2,4]    -> prints the mismatching (config, kind, seed) list"""
import sys
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import emulib as E
import oraclelib as O

F = 6
KINDS = ("dc", "nyquist", "impulse", "bin_sine", "lsb", "two_sample", "sparse", "dc_step", "square_bin", "half_silent")


def make(kind, seed):
    """-> int16 [F, 2, 1152]"""
    rng = np.random.default_rng(seed)
    n = F * 1152
    t = np.arange(n)
    out = np.zeros((2, n), dtype=np.int64)
    for ch in range(2):
        a = int(rng.choice([1, 2, 100, 4000, 32767]))
        if kind == "dc":
            v = np.full(n, int(rng.choice([a, -a, -32768])))
        elif kind == "nyquist":
            v = np.where(t % 2 == 0, a, -a)
        elif kind == "impulse":
            v = np.zeros(n, dtype=np.int64); v[int(rng.integers(n))] = int(rng.choice([a, -a, -32768])); v[int(rng.integers(n))] += int(rng.choice([0, 1, -1]))
        elif kind == "bin_sine":
            p = int(2 ** rng.integers(2, 10)); ph = int(rng.integers(p))
            v = np.round(a * np.sin(2 * np.pi * ((t + ph) % p) / p)).astype(np.int64)      # numpy's sin only shapes the TEST signal
        elif kind == "lsb":
            v = rng.integers(-1, 2, n)
        elif kind == "two_sample":
            p = int(rng.integers(2, 40)); v = np.zeros(n, dtype=np.int64); v[::p] = a; v[1::p] = -a if rng.random() < 0.5 else a
        elif kind == "sparse":
            v = np.zeros(n, dtype=np.int64); idx = rng.integers(0, n, int(rng.integers(1, 12))); v[idx] = rng.integers(-32768, 32768, len(idx))
        elif kind == "dc_step":
            k = int(rng.integers(n)); v = np.where(t < k, a, int(rng.choice([-a, 0, a // 2])))
        elif kind == "square_bin":
            p = int(2 ** rng.integers(1, 9)); v = np.where((t // p) % 2 == 0, a, -a)
        else:                                           # half_silent: noise that stops / starts inside a frame
            k = int(rng.integers(n)); v = rng.integers(-a, a + 1, n); v[k:] = 0 if rng.random() < 0.5 else v[k:]
            if rng.random() < 0.5: v[:k] = 0
        out[ch] = np.clip(v, -32768, 32767)
        if rng.random() < 0.3: out[1] = out[0]; break
    return np.ascontiguousarray(out.reshape(2, F, 1152).transpose(1, 0, 2)).astype(np.int16)


CONFIGS = [(48000, "s", 192), (48000, "j", 128), (48000, "m", 64), (44100, "s", 384), (32000, "j", 192), (24000, "s", 64), (16000, "m", 24), (22050, "j", 160)]


def job(a):
    fs, mode, kbps, psy, kind, seed = a
    pcm = make(kind, seed)
    ref = O.oracle_stream(pcm, samplerate=fs, mode=mode, kbps=kbps, psy=psy)[0]
    b = E.EmuBatch([dict(samplerate=fs, mode=mode, kbps=kbps, psy=psy)])
    g1, _ = b.encode(pcm[:1, None]); g2, _ = b.encode(pcm[1:, None]); tail = b.flush(); b.close()
    return a if g1[0] + g2[0] + tail[0] != ref else None


def run(per_kind=40, seed=1, models=(2, 4), workers=8):
    rng = np.random.default_rng(seed)
    jobs = []
    for kind in KINDS:
        for _ in range(per_kind):
            fs, mode, kbps = CONFIGS[rng.integers(len(CONFIGS))]
            jobs.append((fs, mode, kbps, int(rng.choice(models)), kind, int(rng.integers(1 << 30))))
    E.lib(); O.lib()
    with ProcessPoolExecutor(workers) as ex:
        bad = [r for r in ex.map(job, jobs, chunksize=8) if r]
    return len(jobs), bad


if __name__ == "__main__":
    models = tuple(int(m) for m in sys.argv[3].split(",")) if len(sys.argv) > 3 else (2, 4)
    n, bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1, models)
    print(f"{n} edge streams x {F} frames, psy {models}, emulation vs oracle: {len(bad)} mismatching", bad[:10])
    sys.exit(1 if bad else 0)
