#!/usr/bin/env python3
"""Tone labelling, scalar walk (one iteration per candidate; TLB_LIB_PATH=build/lib_scalarwalk.so) against the lane-parallel rounds (one per confirmed
tone): kernel time per frame of 4096 psy-1 / psy-3 stereo streams on (a) the bench signal (3 tones + noise: ~13 candidates per channel), (b) dense tonal
spectra (tools/fuzz_emu.py crafted(): up to 40 tones per channel at the run-length boundaries), (c) white noise.  usage: tools/walk_ab.py"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
import odr_audioenc_amd as M
from fuzz_emu import crafted
from pcmgen import gen_pcm

S, F = 4096, 12
sigs = {"bench signal (3 tones + noise)": lambda s: gen_pcm(s, 0, 0, F), "dense tones (crafted)": lambda s: np.concatenate([crafted(s), crafted(s + 1000)])[:F],
        "white noise": lambda s: gen_pcm(s, 4, 0, F)}
for name, gen in sigs.items():
    base = np.stack([gen(s) for s in range(64)], axis=1)
    pcm = torch.from_numpy(np.tile(base, (1, S // 64, 1, 1)).copy()).cuda()
    for psy in (1, 3):
        b = M.Batch([M.StreamConfig(mode="s", psy_model=psy)] * S)
        out = torch.zeros((F, S, b.out_stride), dtype=torch.uint8, device="cuda")
        for _ in range(3):
            b.encode_device(pcm.data_ptr(), F, out.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            b.encode_device(pcm.data_ptr(), F, out.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f"{name:32s} psy {psy}: {S * F / dt / 1e6:7.2f} M frames/s")
        b.close()
