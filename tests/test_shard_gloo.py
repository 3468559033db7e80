"""The N>1 path on CPU: world_size-2 `gloo` processes.  Each rank encodes its shard of the streams with
the TEST-ONLY emulation of the kernel (no GPU here); the union of the shards must equal the
single-process result byte for byte (streams share no state), and the two scalars the benchmark
exchanges (max elapsed, total frames) must reduce correctly."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent

WORKER = r"""
import os, sys, pickle, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'tests'))
import odr_audioenc_amd.shard as shard
import emulib as E
from pcmgen import gen_pcm
rank, local_rank, world, dist = shard.init_from_env('gloo')
S, NF = 3, 4                                   # streams per rank, frames
ids = list(shard.weak_stream_ids(rank, S))
pcm = np.stack([gen_pcm(i, i % 8, 0, NF) for i in ids], axis=1)
out = {}
def run():
    b = E.EmuBatch([dict(mode='j', psy=1)] * S)
    got, _ = b.encode(pcm)
    tail = b.flush()
    for k, i in enumerate(ids): out[i] = got[k] + tail[k]
elapsed = shard.timed_region(dist, run)
mx = shard.reduce_max(dist, 10.0 + rank)
total = shard.reduce_sum(dist, S * NF)
lo, hi = shard.strong_range(rank, world, 7)
pickle.dump(dict(rank=rank, world=world, out=out, mx=mx, total=total, rng=(lo, hi), elapsed=elapsed), open(sys.argv[2] + str(rank), 'wb'))
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_sharding(tmp_path):
    import pickle
    sys.path.insert(0, str(ROOT / "tests"))
    import emulib as E
    from pcmgen import gen_pcm
    E.lib()                                    # build once, before the workers race for it
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER, str(ROOT), str(tmp_path / "res")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        outp, _ = p.communicate(timeout=240)
        assert p.returncode == 0, outp.decode()[-2000:]
    res = [pickle.load(open(str(tmp_path / "res") + str(r), "rb")) for r in range(2)]
    assert [r["mx"] for r in res] == [11.0, 11.0]
    assert [r["total"] for r in res] == [24, 24]
    assert res[0]["rng"] == (0, 3) and res[1]["rng"] == (3, 7)
    assert res[0]["elapsed"] == res[1]["elapsed"] > 0
    merged = {**res[0]["out"], **res[1]["out"]}
    assert sorted(merged) == list(range(6))
    # single-process run over all six streams
    pcm = np.stack([gen_pcm(i, i % 8, 0, 4) for i in range(6)], axis=1)
    b = E.EmuBatch([dict(mode="j", psy=1)] * 6)
    got, _ = b.encode(pcm)
    tail = b.flush()
    for i in range(6):
        assert merged[i] == got[i] + tail[i]
