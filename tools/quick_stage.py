#!/usr/bin/env python3
"""Per-kernel time of the split path (psy kernel / encode kernel) on the GPU box: tools/quick_stage.py [streams] [frames] [psy] [mode]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import odr_audioenc_amd as M
from pcmgen import gen_pcm
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
psy = int(sys.argv[3]) if len(sys.argv) > 3 else 1
mode = sys.argv[4] if len(sys.argv) > 4 else "s"
base = np.stack([gen_pcm(s, 0, 0, 2 * F) for s in range(256)], axis=1)
host = np.tile(base, (1, S // 256 + 1, 1, 1))[:, :S]
pcm = [torch.from_numpy(host[:F].copy()).cuda(), torch.from_numpy(host[F:].copy()).cuda()]
b = M.Batch([M.StreamConfig(mode=mode, psy_model=psy)] * S)
out = torch.zeros((F, S, b.out_stride), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream()
res = []
for i in range(12):
    b.encode_device(pcm[i & 1].data_ptr(), F, out.data_ptr(), stream=st.cuda_stream)
    tot = b.last_kernel_ms()
    res.append((tot,) + (b.last_stage_ms() or (0.0, 0.0)))
r = np.array(res[4:])
print(f"S={S} F={F} psy={psy} mode={mode}: total {r[:,0].mean():.4f} ms  psy {r[:,1].mean():.4f} ms  encode {r[:,2].mean():.4f} ms  -> {S*F/r[:,0].mean()/1e3:.2f} M frames/s")
