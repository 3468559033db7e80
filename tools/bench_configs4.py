#!/usr/bin/env python3
"""One GPU's share of BASELINE configs[4] on the GPU box: 16384 streams, even = 32 kHz mono 64 kbps, odd = 48 kHz stereo
192 kbps, interleaved in one batch, psy 4 (and psy 2), 8 frames per stream per launch.   usage: tools/bench_configs4.py [psy]"""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import odr_audioenc_amd as M
from pcmgen import gen_pcm

psy = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S, F, steps = 16384, 8, 20
cfgs = [M.StreamConfig(samplerate=32000, mode="m", bitrate=64, psy_model=psy) if s % 2 == 0 else
        M.StreamConfig(samplerate=48000, mode="s", bitrate=192, psy_model=psy) for s in range(S)]
base = np.stack([gen_pcm(s, 0, 0, 2 * F) for s in range(256)], axis=1)
host = np.tile(base, (1, S // 256, 1, 1))
pcm = [torch.from_numpy(host[:F].copy()).cuda(), torch.from_numpy(host[F:].copy()).cuda()]
b = M.Batch(cfgs)
out = torch.zeros((F, S, b.out_stride), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream()
for i in range(3):
    b.encode_device(pcm[i & 1].data_ptr(), F, out.data_ptr(), stream=st.cuda_stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    b.encode_device(pcm[i & 1].data_ptr(), F, out.data_ptr(), stream=st.cuda_stream)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
frames = S * F * steps
# a 32 kHz frame is 36 ms of audio, a 48 kHz frame 24 ms: seconds of audio encoded per second
audio_s = (S // 2) * F * steps * (1152 / 32000 + 1152 / 48000)
print(json.dumps({"workload": f"{S} streams, even 32 kHz mono 64 kbps / odd 48 kHz stereo 192 kbps, psy {psy}, {F} frames/stream/launch (BASELINE configs[4], one GPU's share)",
                  "frames_per_s": round(frames / dt, 1), "ms_per_launch": round(dt / steps * 1e3, 3),
                  "realtime_factor_per_stream": round(audio_s / dt / S, 1), "streams_at_realtime": round(audio_s / dt)}))
b.close()
