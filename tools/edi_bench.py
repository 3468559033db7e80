#!/usr/bin/env python3
"""Diagnostic: time of the EDI AF-packet kernel next to the encode kernel (run on the GPU box)."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import odr_audioenc_amd as M

S, F = 4096, 8
b = M.Batch([M.StreamConfig()] * S)
L = M.load_library()
ver = b"v3.5.0-graft"
ps = L.tlb_edi_af_stride(b.h, len(ver))
dev = torch.device("cuda:0")
frames = torch.randint(0, 256, (F, S, b.out_stride), dtype=torch.uint8, device=dev)
levels = torch.zeros((F, S, 2), dtype=torch.int16, device=dev)
st = torch.from_numpy(M.edi_state_init(S, 1700000000, 0, True, 37).view(np.uint8).reshape(S, -1)).to(dev)
pkts = torch.zeros((F, S, ps), dtype=torch.uint8, device=dev)
plen = torch.zeros((F, S), dtype=torch.int32, device=dev)
for _ in range(3):
    assert L.tlb_edi_af_device(b.h, frames.data_ptr(), levels.data_ptr(), F, st.data_ptr(), ver, len(ver), pkts.data_ptr(), plen.data_ptr(), None) == 0
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 20
for _ in range(N):
    L.tlb_edi_af_device(b.h, frames.data_ptr(), levels.data_ptr(), F, st.data_ptr(), ver, len(ver), pkts.data_ptr(), plen.data_ptr(), None)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print(f"EDI AF packets: {S} streams x {F} frames in {dt * 1e3:.3f} ms = {S * F / dt / 1e6:.1f} M packets/s "
      f"({int(plen[0, 0])} B each, {S * F * int(plen[0, 0]) / dt / 1e9:.1f} GB/s written)")

# PFT layer on the AF packets just built
import ctypes as C
for fec in (0, 2):
    mf, fs = C.c_int(0), C.c_int(0)
    assert L.tlb_edi_pft_shape(b.h, ps, fec, 207, 0, C.byref(mf), C.byref(fs)) == 0
    frags = torch.zeros((F, S, mf.value, fs.value), dtype=torch.uint8, device=dev)
    flen = torch.zeros((F, S, mf.value), dtype=torch.int32, device=dev)
    nfr = torch.zeros((F, S), dtype=torch.int32, device=dev)
    pseq = torch.zeros((S,), dtype=torch.int16, device=dev)
    args = (b.h, pkts.data_ptr(), plen.data_ptr(), F, ps, pseq.data_ptr(), fec, 207, 0, 0, 0, frags.data_ptr(), flen.data_ptr(), nfr.data_ptr(), mf.value, fs.value, None)
    for _ in range(3):
        assert L.tlb_edi_pft_device(*args) == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        L.tlb_edi_pft_device(*args)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    print(f"EDI PFT fec={fec}: {S * F} AF packets -> {int(nfr[0, 0])} fragment(s) each in {dt * 1e3:.3f} ms = {S * F / dt / 1e6:.1f} M packets/s")
