#!/usr/bin/env python3
"""Diagnostic: per-stage share of a frame's cycles inside the encode and psy kernels (s_memtime stamps).
Run on the GPU box: python tools/stage_profile.py [psy] [mode] [nstreams]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import odr_audioenc_amd as M
from pcmgen import gen_pcm

psy = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mode = sys.argv[2] if len(sys.argv) > 2 else "s"
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
F = 4
pcm = np.stack([gen_pcm(s, 0, 0, F) for s in range(S)], axis=1)
b = M.Batch([M.StreamConfig(mode=mode, psy_model=psy)] * S)
L = M.load_library()
L.tlb_encode_host_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
st = np.zeros((F, S, 32), dtype=np.int64)
assert L.tlb_encode_host_stamps(b.h, np.ascontiguousarray(pcm).ctypes.data, F, st.ctypes.data) == 0
st = st[1:]                                   # skip the first (cold) frame
names = ["filterbank", "scalefactors", "psy 0" if psy == 0 else "smr from the model's record", "sfpattern+bitalloc", "quantise+pack", "crc+scfcrc+pad", "emit"]
tot = (st[..., 7] - st[..., 0]).mean()
print(f"psy {psy} mode {mode} {S} streams: {tot:.0f} s_memtime ticks/frame/wave in the encode kernel")
if psy in (1, 3):
    ptot = (st[..., 23] - st[..., 15]).mean()
    print(f"psy phase: {ptot:.0f} ticks/unit/wave; shares below are of encode + psy = {tot + ptot:.0f}")
    tot = tot + ptot
for i, n in enumerate(names):
    d = (st[..., i + 1] - st[..., i]).mean()
    print(f"  {n:22s} {d:10.0f}  {100 * d / tot:5.1f}%")
pn = ["spectrum(FHT)", "power+candidates", "tonal walk", "noise bands", "decimation", "threshold", "minmask+smr"]
if psy in (2, 4):      # stamps of the first of the two 576-sample passes of a frame, in tl_psy2_kernel (shares: of the ENCODE kernel's ticks)
    pn = ["window+FHT", "energy/phase/unpred.", "partitions", "spreading+SNR", "line thresholds", "subbands", "-"]
for ch in range(2):
    base = 8 + 8 * ch
    if st[..., base + 1].max() == 0:
        continue
    for i, n in enumerate(pn):
        hi = st[..., base + i + 1] if i < 6 else None
        if hi is None:
            continue
        d = (hi - st[..., base + i]).mean()
        print(f"    ch{ch} {n:18s} {d:10.0f}  {100 * d / tot:5.1f}%")
if st[..., 25].max() > 0:
    fn = ["window+scatter", "first pass", "pass k=2", "pass k=4", "pass k=6", "pass k=8"]
    for i, n in enumerate(fn):
        d = (st[..., 25 + i] - st[..., 24 + i]).mean()
        print(f"      ch0 FHT {n:16s} {d:8.0f}  {100 * d / tot:5.1f}%")
if st[..., 31].max() > 0:
    d = (st[..., 0] - st[..., 31]).mean()
    print(f"  PCM staging before stamp 0: {d:8.0f}  {100 * d / tot:5.1f}% (not in the total above)")
if psy in (1, 3) and st[..., 31].max() > 0:
    # the pieces of a unit outside the stage stamps: unit begin -> the first channel's spectrum (launch record, configuration record), model end ->
    # encoder begin; what is then still missing to (kernel time x clock / units per wave) is taking the unit: the atomic, stream_list[k], the loop
    a = (st[..., 8] - st[..., 15]).mean(); bgap = (st[..., 31] - st[..., 23]).mean()
    whole = (st[..., 7] - st[..., 15]).mean()
    print(f"  unit begin -> ch0 spectrum: {a:8.0f}   model end -> encoder begin: {bgap:8.0f}   unit begin -> emit done: {whole:8.0f} ticks")

