"""Node level on the GPU (include/toolame_batch.h part 3, csrc/tlb_node.cpp): N streams cut into contiguous blocks, one shard
(tlb_tick or tlb_batch + its own host thread) per block.  The test box has ONE MI355X, so the shards are devices = (0, 0[, 0]):
the partition, the threads, the per-shard objects and the aggregation are exactly those of an 8-GPU node, only the device ordinal
repeats.  The bar: stream for stream the bytes of ONE single batch and of the oracle."""
import threading
import time

import numpy as np
import pytest

import oraclelib as O
from pcmgen import gen_pcm

pytestmark = pytest.mark.gpu

MIX = [(48000, "s", 128, 1), (48000, "j", 128, 3), (24000, "m", 64, 1), (48000, "s", 192, 2), (48000, "m", 96, 4), (16000, "m", 32, 3),
       (48000, "m", 64, 0), (48000, "m", 96, 4), (48000, "j", 192, 2), (24000, "m", 64, 1), (48000, "s", 128, 1), (44100, "s", 128, 1),
       (48000, "m", 64, 0)]


@pytest.fixture(scope="module")
def M():
    import odr_audioenc_amd as mod
    mod.load_library()
    return mod


def _cfgs(M, streams):
    return [M.StreamConfig(samplerate=r, mode=m, bitrate=k, psy_model=p) for r, m, k, p in streams]


def _planar(inter_s, c):
    """what the ingest stage makes of a stream's interleaved frames [T][2304]: L R L R ... -> [T][2][1152]; a MONO stream is its first
    1152 values (include/toolame_batch.h, tlb_ingest_device)"""
    T = inter_s.shape[0]
    if c.mode == "m":
        return np.repeat(inter_s[:, None, :1152], 2, axis=1)
    return inter_s.reshape(T, 1152, 2).transpose(0, 2, 1)


def _oracle(pcm_s, c):
    return O.oracle_stream(pcm_s, samplerate=c.samplerate, mode=c.mode, kbps=c.bitrate, psy=c.psy_model)[0]


@pytest.mark.parametrize("devices", [(0, 0), (0, 0, 0)])
def test_batch_plane_two_shards_equal_one_batch_and_the_oracle(M, devices):
    """13 mixed streams (every psy model, mono pairs that the cut separates, an LSF rate, a padded rate) over 2 and 3 shards on the
    one GPU, two ragged calls + flush: the node's bytes = a single tlb_batch's = the oracle's, stream for stream."""
    cfgs = _cfgs(M, MIX)
    ns, nf = len(cfgs), 7
    pcm = np.stack([gen_pcm(4100 + s, (0, 7, 5, 4)[s % 4], 0, nf) for s in range(ns)], axis=1)
    one = M.Batch(cfgs)
    a1, _ = one.encode(pcm[:3])
    a2, _ = one.encode(pcm[3:])
    at = one.flush()
    one.close()
    nd = M.Node(cfgs, devices=devices, plane="batch")
    assert nd.blocks == M.node_partition(ns, len(devices))
    b1 = nd.encode(pcm[:3])
    b2 = nd.encode(pcm[3:])
    bt = nd.flush()
    per, tot = nd.counters()
    nd.close()
    for s in range(ns):
        assert b1[s] == a1[s] and b2[s] == a2[s] and bt[s] == at[s], s
        assert b1[s] + b2[s] + bt[s] == _oracle(pcm[:, s], cfgs[s]), s
    assert tot["frames"] == ns * nf and [p["frames"] for p in per] == [n * nf for _, n in M.node_partition(ns, len(devices))]
    assert all(p["steps"] == 2 and p["busy_ns"] > 0 and p["device"] == 0 for p in per) and tot["wall_ns"] > 0


@pytest.mark.parametrize("egress", ["frames", "af", "zmq"])
def test_tick_plane_two_shards_equal_one_tick(M, egress):
    """The real-time loop through the node, ticks overlapped (submit, submit, wait ...): every tick, every stream, frame / packets /
    messages / levels / silence counter equal ONE tlb_tick over all streams; raw frames also equal the oracle."""
    streams = [(48000, "s", 128, 1), (48000, "j", 128, 3), (24000, "m", 64, 1), (48000, "s", 192, 2), (48000, "m", 96, 4), (48000, "m", 96, 4), (16000, "m", 32, 3)]
    if egress == "frames":
        streams = streams + [(32000, "m", 64, 1), (44100, "j", 128, 3)]             # rates only raw frames can carry
    cfgs = _cfgs(M, streams)
    ns, T = len(cfgs), 9
    inter = np.stack([np.stack([gen_pcm(5200 + s, (0, 7, 5, 4)[s % 4], 0, T)[f].T.reshape(-1) for s in range(ns)]) for f in range(T)])
    kw = dict(egress=egress, version=b"v5", now_s=1712345678, delay_ms=370, tist=True)
    snap = lambda t, per_stream: [(t.frame(s), t.packets(s), t.messages(s), tuple(int(v) for v in (t.peaks[s] if per_stream else t.peaks(s))),
                                   int(t.silence_ms[s]) if per_stream else t.silence_ms(s)) for s in range(ns)]
    a = M.Tick(cfgs, ngroups=2, **kw)
    want = []
    for f in range(T):
        a.pcm[:] = inter[f]
        a.run()
        want.append(snap(a, True))
    a.finish()
    want.append(snap(a, True))
    a.close()
    nd = M.Node(cfgs, devices=(0, 0), plane="tick", ngroups=1, **kw)
    got = []
    nd.set_pcm(inter[0])
    nd.submit()
    for f in range(1, T):
        nd.set_pcm(inter[f])
        nd.submit()
        assert nd.pcm(0) is None and nd.pcm(ns - 1) is None      # two ticks in flight: no input set is free
        nd.wait()
        got.append(snap(nd, False))
    nd.wait()
    got.append(snap(nd, False))
    nd.finish()
    got.append(snap(nd, False))
    per, tot = nd.counters()
    nd.close()
    assert len(got) == len(want) == T + 1
    for f in range(T + 1):
        for s in range(ns):
            assert got[f][s] == want[f][s], (f, s)
    assert tot["frames"] == ns * T and all(p["steps"] == T for p in per)
    if egress == "frames":
        for s in range(ns):
            assert b"".join(got[f][s][0] for f in range(T + 1)) == _oracle(_planar(inter[:, s], cfgs[s]), cfgs[s]), s


def test_tick_results_stay_valid_until_the_next_wait(M):
    """with two ticks in flight the results of the tick waited for last must survive the submit after the next
    one.  s0 s1 w0 [read] s2 [read again: unchanged] w1 ..."""
    cfgs = _cfgs(M, [(48000, "s", 128, 1)] * 6)
    ns, T = len(cfgs), 6
    inter = np.stack([np.stack([gen_pcm(6100 + s, (0, 7)[s % 2], 0, T)[f].T.reshape(-1) for s in range(ns)]) for f in range(T)])
    t = M.Tick(cfgs, egress="af", ngroups=2, version=b"x")
    snap = lambda: [(t.packets(s), tuple(int(v) for v in t.peaks[s])) for s in range(ns)]
    t.pcm[:] = inter[0]; t.submit()
    t.pcm[:] = inter[1]; t.submit()
    assert t.pcm is None
    for f in range(2, T):
        t.wait()
        before = snap()
        t.pcm[:] = inter[f]; t.submit()
        time.sleep(0.05)                                         # the submitted tick's kernels and copy-out land meanwhile
        assert snap() == before, f
    t.wait(); t.wait()
    t.close()


def test_node_life_cycle_and_gain_route_to_the_owning_shard(M):
    cfgs = _cfgs(M, [(48000, "s", 128, 1), (48000, "j", 128, 3), (48000, "s", 192, 2), (48000, "m", 96, 4), (48000, "m", 96, 4)])
    ns, T = len(cfgs), 6
    inter = np.stack([gen_pcm(7300 + s, 0, 0, T) for s in range(ns)], axis=1).transpose(0, 1, 3, 2).reshape(T, ns, 2304)
    pcm = np.stack([_planar(inter[:, s], cfgs[s]) for s in range(ns)], axis=1)
    nd = M.Node(cfgs, devices=(0, 0), plane="tick", egress="frames")
    got = [b""] * ns
    for f in range(3):
        nd.set_pcm(inter[f]); nd.run()
        for s in range(ns):
            got[s] += nd.frame(s)
    last3 = nd.stream_finish(3)                                   # stream 3 lives in shard 1 (block [2, 5))
    assert nd.L.tlb_node_shard_of(nd.h, 3) == 1 and nd.L.tlb_node_shard_of(nd.h, 1) == 0
    assert got[3] + last3 == _oracle(pcm[:3, 3], cfgs[3])
    nd.stream_reconfigure(1, M.StreamConfig(mode="s", bitrate=128, psy_model=1))
    after = {1: b"", 3: b""}
    for f in range(3, T):
        nd.set_pcm(inter[f]); nd.run()
        for s in range(ns):
            if s in after:
                after[s] += nd.frame(s)
            else:
                got[s] += nd.frame(s)
    nd.finish()
    for s in range(ns):
        if s in after:
            after[s] += nd.frame(s)
        else:
            got[s] += nd.frame(s)
    nd.close()
    for s in (0, 2, 4):
        assert got[s] == _oracle(pcm[:, s], cfgs[s]), s
    assert after[3] == _oracle(pcm[3:, 3], cfgs[3])
    assert after[1] == _oracle(pcm[3:, 1], M.StreamConfig(mode="s", bitrate=128, psy_model=1))


def test_node_parallel_runs_on_every_shard_thread(M):
    import ctypes as C
    cfgs = _cfgs(M, [(48000, "s", 128, 1)] * 10)
    nd = M.Node(cfgs, devices=(0, 0, 0), plane="tick", egress="frames")
    seen = {}
    CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int)

    def fn(ctx, shard, first, n):
        seen[shard] = (first, n, threading.get_ident())
    cb = CB(fn)
    assert nd.L.tlb_node_parallel(nd.h, cb, None) == 0
    nd.close()
    assert sorted(seen) == [0, 1, 2] and [(seen[g][0], seen[g][1]) for g in range(3)] == M.node_partition(10, 3)
    assert len({seen[g][2] for g in range(3)}) == 3 and threading.get_ident() not in {seen[g][2] for g in range(3)}


def test_batch_plane_device_clock_sums_every_queued_step(M):
    """the BATCH plane's device_ms is the device time of EVERY completed step: each queued encode call has its own pair of events on
    the shard's stream (include/toolame_batch.h, tlb_node_counter.device_ms).  Four calls queued behind ONE sync cost about four times
    one call's device time -- with only the last launch's figure (round 5) the two would read the same."""
    cfgs = _cfgs(M, [(48000, "s", 128, 1)] * 512)
    pcm = np.stack([gen_pcm(77 + (s & 7), 0, 0, 4) for s in range(512)], axis=1)
    nd = M.Node(cfgs, devices=(0, 0), plane="batch")
    nd.upload(pcm)
    nd.encode_resident(); nd.sync()                                 # warm (tables, first-launch costs)
    base = nd.counters()[1]["device_ms"]
    nd.encode_resident(); nd.sync()
    one = nd.counters()[1]["device_ms"] - base
    base += one
    for _ in range(4):
        nd.encode_resident()
    nd.sync()
    per, tot = nd.counters()
    four = tot["device_ms"] - base
    nd.close()
    assert one > 0 and 2.5 * one < four < 8 * one, (one, four)
    assert all(p["steps"] == 6 and p["frames"] == 6 * 4 * 256 for p in per)


def _bench(*args, timeout=900):
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "bench.py")] + list(args), cwd=root, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.strip().splitlines()
    assert lines[-1].startswith("{"), lines[-3:]                    # the JSON line is the LAST thing on stdout (librccl's own printf comes before it)
    return json.loads(lines[-1])


def test_bench_forced_rccl_group_on_one_gpu():
    """the `nccl` branch of the harness -- init_process_group("nccl", device_id=...), barrier, all_reduce(MAX),
    all_gather on DEVICE tensors -- executed through RCCL on the MI355X, as a one-rank group (`--force-group`).  An 8-GPU driver
    run then repeats with more ranks what has already run once."""
    line = _bench("--gpus", "1", "--backend", "nccl", "--force-group", "--steps", "3", "--warmup", "1", "--no-also", "--no-cpu-baseline")
    assert line["collective_backend"].startswith("rccl") and line["world_size_observed"] == 1 and line["n_gpus"] == 1
    assert line["collectives_executed"] == ["init_process_group(device_id)", "barrier", "all_reduce(MAX)", "all_gather"]
    assert line["value"] > 0 and line["output_check"]["checked"] and line["output_check"]["per_rank_ok"] == [True]
    rf = line["roofline"]
    assert rf["bound"] == "valu_fp64" and rf["unit"] == "TFLOP/s" and rf["peak"] == 78.6 and 0 < rf["frac"] < 0.5
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0 < rf["hbm_frac"] < 0.1
    assert "valu_issue_utilisation" in rf and "NOT a roofline fraction" in rf["valu_issue_utilisation"]["what"]
    # the N = 1 line carries the device's identity too (gathered over RCCL here), and the clock probe names the card it read by PCI address
    dev = line["devices_observed"]
    print("devices_observed:", dev)
    assert len(dev) == 1 and line["distinct_devices_observed"] == 1 and dev[0]["name"] and (dev[0].get("pci_bus_id") or dev[0].get("uuid")), dev
    assert dev[0]["compute_units"] == 256


def test_bench_in_process_node_two_shards_on_the_gpu():
    """`bench.py --in-process 2`: the product's node object (two shards, two host threads, both on this box's GPU) encodes 2 x 4096
    psy-1 streams; the node's own counters agree with the harness's count and the oracle check passes on the first and last stream."""
    line = _bench("--in-process", "2", "--steps", "3", "--warmup", "1", "--no-also", "--no-cpu-baseline")
    nd = line["node"]
    assert nd["shards"] == 2 and nd["devices"] == [0, 0] and nd["value"] > 0 and nd["output_check"]["checked"]
    assert nd["frames_counted_by_the_node"] == 2 * 4096 * 32 * 3
    assert [p["nstreams"] for p in nd["per_shard"]] == [4096, 4096] and [p["first"] for p in nd["per_shard"]] == [0, 4096]
    print(nd["describe"])
    assert nd["describe"].count("shard ") == 2 and "256 CUs in 8 XCDs" in nd["describe"] and "uuid" in nd["describe"]
