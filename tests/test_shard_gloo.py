"""The N>1 path on CPU: world_size-2 `gloo` processes.  Each rank encodes its shard of the streams with
the TEST-ONLY emulation of the kernel (no GPU here); the union of the shards must equal the
single-process result byte for byte (streams share no state), and the two scalars the benchmark
exchanges (max elapsed, total frames) must reduce correctly."""
import os
import pickle
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


def _run_bench(args, timeout=300):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")})
    return r


def test_bench_gpus_spawns_ranks_dry_run():
    """`python bench.py --gpus 2` (no launcher) starts two rank processes itself; with --dry-run the emulation stands in for
    the GPU.  The line says n_gpus 2, the process group saw two ranks, both ranks contributed frames."""
    import json
    r = _run_bench(["--gpus", "2", "--dry-run", "--steps", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["world_size_observed"] == 2 and line["dry_run"] is True
    assert len(line["per_gpu_frames_per_s"]) == 2 and all(v > 0 for v in line["per_gpu_frames_per_s"])
    assert "configs[3]" not in line["config"]["workload"]          # 2 emulated streams per rank are not the BASELINE config
    # the same code path at N = 1
    r1 = _run_bench(["--gpus", "1", "--dry-run", "--steps", "2"])
    assert r1.returncode == 0, r1.stderr[-2000:]
    l1 = json.loads(r1.stdout.strip().splitlines()[-1])
    assert l1["n_gpus"] == 1 and l1["world_size_observed"] == 1 and len(l1["per_gpu_frames_per_s"]) == 1


def test_bench_eight_ranks_dry_run_names_eight_devices():
    """The driver's largest world size through the same plumbing: eight rank processes, one process group, and the
    line carries `devices_observed` -- one identity record per rank gathered with ONE all_gather over the run's own backend -- so that a
    record of a real 8-GPU run proves by itself that N ranks ran on N distinct devices."""
    import json
    r = _run_bench(["--gpus", "8", "--dry-run", "--steps", "1"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["world_size_observed"] == 8 and len(line["per_gpu_frames_per_s"]) == 8
    dev = line["devices_observed"]
    assert [d["rank"] for d in dev] == list(range(8)) and line["distinct_devices_observed"] == 8
    assert len({d["pid"] for d in dev}) == 8 and all("host" in d and "setup_s" in d for d in dev)


def test_gather_bytes_over_gloo(tmp_path):
    """shard.gather_bytes: every rank's byte string comes back on every rank, in rank order, padding stripped"""
    import odr_audioenc_amd.shard as shard
    assert shard.gather_bytes(None, b"alone") == [b"alone"]
    child = r"""
import os, sys, pickle
sys.path.insert(0, sys.argv[1])
import odr_audioenc_amd.shard as shard
rank, lr, world, dist = shard.init_from_env("gloo")
got = shard.gather_bytes(dist, ("gpu-of-rank-%d" % rank).encode() * (rank + 1), width=64)
pickle.dump(got, open(sys.argv[2] + str(rank), "wb"))
dist.destroy_process_group()
"""
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, "-c", child, str(ROOT), str(tmp_path / "gb")],
                              env=dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
             for r in range(2)]
    for p_ in procs:
        assert p_.wait(timeout=240) == 0
    for r in range(2):
        assert pickle.load(open(str(tmp_path / "gb") + str(r), "rb")) == [b"gpu-of-rank-0", b"gpu-of-rank-1gpu-of-rank-1"]


def test_bench_rejects_world_size_mismatch():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--dry-run"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)


def test_bench_workload_labels():
    sys.path.insert(0, str(ROOT))
    import bench
    assert "configs[1]" in bench.workload_label(4096, 1, "s", 16, 1)[0]
    assert "configs[2]" in bench.workload_label(16384, 3, "s", 8, 1)[0]
    lab, k = bench.workload_label(16384, 3, "s", 8, 8)
    assert k == 3 and "configs[3]" in lab and "131072 streams in total" in lab
    assert bench.workload_label(4096, 3, "s", 8, 1)[1] is None
    # the BASELINE configurations are plain stereo -- mono pairs, joint stereo and dual channel are NOT configs[1] / [2]
    for mode in ("m", "j", "d"):
        lab, k = bench.workload_label(4096, 1, mode, 32, 1)
        assert k is None and "not a BASELINE config" in lab and "BASELINE configs[" not in lab, lab
        assert bench.workload_label(16384, 3, mode, 8, 8)[1] is None
    assert "mono, two streams per wave" in bench.workload_label(4096, 1, "m", 32, 1)[0]
    a = bench.parse_args(["--gpus", "8"])
    assert a.gpus == 8 and a.streams is None and a.psy is None


def test_forced_one_rank_group_runs_the_collectives():
    """`--force-group`: at world size 1 the harness still builds a process group, so barrier / all_reduce / all_gather run
    through the backend's code -- gloo here, RCCL on the GPU box (tests/test_node_gpu.py::test_bench_forced_rccl_group_on_one_gpu)."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--dry-run", "--force-group", "--steps", "1"], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["collective_backend"] == "gloo" and line["world_size_observed"] == 1 and line["n_gpus"] == 1
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--dry-run", "--steps", "1"], cwd=root, capture_output=True, text=True, timeout=600)
    assert json.loads(r.stdout.strip().splitlines()[-1])["collective_backend"] is None          # without the flag: no group at world size 1
