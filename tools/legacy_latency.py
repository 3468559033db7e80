#!/usr/bin/env python3
"""Per-call time of the drop-in ABI (toolame_encode_frame, one stream, one frame per call) on the GPU box: what an unchanged
odr-audioenc pays every 24 ms.   usage: tools/legacy_latency.py [frames] [psy model 0..3]"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import odr_audioenc_amd as M
from pcmgen import gen_pcm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
psy = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L = M.legacy_api()
L.toolame_set_samplerate.argtypes = [C.c_long]
L.toolame_set_channel_mode.argtypes = [C.c_char]
L.toolame_encode_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
L.toolame_init(); L.toolame_set_samplerate(48000); L.toolame_set_psy_model(psy); L.toolame_set_channel_mode(b"j")
L.toolame_set_bitrate(128); L.toolame_set_pad(0)
pcm = gen_pcm(0, 0, 0, 64)
out = (C.c_ubyte * 4096)()
ts = []
for i in range(n):
    t = time.perf_counter()
    L.toolame_encode_frame(pcm[i & 63].ctypes.data, None, 0, out, 4096)
    ts.append(time.perf_counter() - t)
ts = np.array(ts[32:]) * 1e3
# the shim defers: calls that return nothing only file the frame away, the call on which the reference's 4096-byte buffer
# fills (about one in 10.7 here) encodes all filed frames in one launch -- so the MEAN is what a file-to-file run pays per frame
burst = ts[ts > 10 * np.median(ts)]
print(f"toolame_encode_frame, 48 kHz joint stereo 128 kbps psy {psy}: mean {ts.mean():.4f} ms per call ({1e3 / ts.mean():.0f} frames/s through the "
      f"unchanged ABI), median {np.median(ts):.4f} ms, p99 {np.percentile(ts, 99):.3f} ms, max {ts.max():.3f} ms over {len(ts)} calls; "
      f"{len(burst)} burst calls, mean {burst.mean() if len(burst) else 0:.3f} ms each (a frame is 24 ms of audio)")
