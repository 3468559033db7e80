#!/usr/bin/env python3
"""Stress the emulated kernel (tests/emu) against the oracle on signals built to hit the tone-labelling
edge cases (range boundaries 63/127/255, near-7-dB neighbours, adjacent tones, head erasure).
usage: tools/fuzz_emu.py [nseeds] [psy...]"""
import sys
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import emulib as E
import oraclelib as O
from pcmgen import gen_pcm

NF = 6


def crafted(seed):
    rng = np.random.default_rng(seed)
    n = np.arange(NF * 1152)
    x = np.zeros((2, NF * 1152))
    ntones = rng.integers(2, 40)
    anchors = np.array([2, 3, 4, 60, 61, 62, 63, 64, 66, 124, 126, 127, 128, 130, 133, 250, 254, 255, 256, 262, 268, 280, 400, 487, 495, 499])
    for ch in range(2):
        for _ in range(ntones):
            if rng.random() < 0.6:
                b = float(rng.choice(anchors)) + rng.choice([0, 0, 0.5, -0.25, 0.25])
            else:
                b = rng.uniform(1, 510)
            if rng.random() < 0.5:
                b2 = b + rng.integers(1, 14)          # a partner tone a few bins up
                amp2 = 10 ** rng.uniform(0.5, 4.2)
                x[ch] += amp2 * np.sin(2 * np.pi * b2 * 46.875 * n / 48000 + rng.uniform(0, 6.28))
            amp = 10 ** rng.uniform(0.5, 4.2)
            x[ch] += amp * np.sin(2 * np.pi * b * 46.875 * n / 48000 + rng.uniform(0, 6.28))
        x[ch] += rng.normal(0, 10 ** rng.uniform(-0.5, 3), n.shape)
    x = np.clip(np.round(x), -32768, 32767).astype(np.int16)
    return np.ascontiguousarray(x.reshape(2, NF, 1152).transpose(1, 0, 2))


def work(args):
    seed, psy = args
    bad = []
    for mode, kbps in (("s", 128), ("j", 192)):
        pcm = crafted(seed) if seed % 2 == 0 else gen_pcm(seed, seed % 8 if not (psy == 3 and seed % 8 in (1, 3)) else 0, 0, NF)
        ref, _ = O.oracle_stream(pcm, mode=mode, kbps=kbps, psy=psy)
        b = E.EmuBatch([dict(mode=mode, kbps=kbps, psy=psy)])
        got, _ = b.encode(pcm[:, None])
        got = got[0] + b.flush()[0]
        b.close()
        if got != ref:
            bad.append((seed, psy, mode))
    return bad


if __name__ == "__main__":
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    psys = [int(a) for a in sys.argv[2:]] or [1, 3]
    E.lib(); O.lib()
    jobs = [(s, p) for s in range(nseeds) for p in psys]
    with ProcessPoolExecutor(8) as ex:
        res = [b for r in ex.map(work, jobs, chunksize=8) for b in r]
    print(len(jobs) * 2 * NF, "frames,", len(res), "mismatching streams", res[:20])
