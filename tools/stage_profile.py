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
def rows_per_channel():
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


if psy in (1, 3) and mode != "m":
    # Two-channel frames (tl_psy1_stereo / tl_psy3_stereo): channel 0 is PARKED after its front while channel 1's front uses the LDS arrays,
    # then BOTH channels' dB-sum chains run side by side, then channel 1's back, then channel 0's.  The kernel stamps the chain stage itself
    # (channel 1's slot 4 = chains begin, channel 0's slot 4 = chains end), so the rows below are the stages in the order they RUN -- a
    # channel's waiting time is a row of its own, never part of its "noise bands" (round 5's table booked channel 1's
    # front, both chains and channel 1's back under "ch0 noise bands").  (The rare dead-head frames take the plain per-channel order and
    # blur the means slightly.)
    c0, c1 = 8, 16
    seq = [("ch0 spectrum (window + FHT)", c0 + 0, c0 + 1), ("ch0 power + candidates", c0 + 1, c0 + 2), ("ch0 tone walk", c0 + 2, c0 + 3),
           ("ch0 compaction + weights, parked", c0 + 3, c1 + 0),
           ("ch1 spectrum (window + FHT)", c1 + 0, c1 + 1), ("ch1 power + candidates", c1 + 1, c1 + 2), ("ch1 tone walk", c1 + 2, c1 + 3),
           ("ch1 compaction + weights", c1 + 3, c1 + 4),
           ("dB-sum chains, BOTH channels", c1 + 4, c0 + 4),
           ("ch1 centres + decimation", c0 + 4, c1 + 5), ("ch1 thresholds", c1 + 5, c1 + 6),
           ("ch0 resumed: centres + decimation", c1 + 6, c0 + 5), ("ch0 thresholds", c0 + 5, c0 + 6)]
    print("  model phase in running order (ticks per unit, share of encode + psy):")
    for n, a, b_ in seq:
        d = (st[..., b_] - st[..., a]).mean()
        print(f"    {n:36s} {d:10.0f}  {100 * d / tot:5.1f}%")
else:
    rows_per_channel()
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

