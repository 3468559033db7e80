"""Fault isolation of the node level on the GPU (include/toolame_batch.h, FAULT ISOLATION; csrc/tlb_node.cpp).

The reference restarts ONE failed input and nothing else (src/odr-audioenc.cpp:875-902, src/InputInterface.h:40); one level up the unit
of failure is a GPU.  A healthy GPU takes none of these paths, so the tests run on the fault-injection TEST build of the library
(odr-audioenc_amd/libtoolame_dab_hip_fi.so: the product's kernel objects + the host files compiled with -DTLB_FAULT_INJECT,
csrc/tlb_debug.h) -- an armed launch fails exactly where a failing device call inside it would.  The bar: with
devices = (0, 0, 0) and a fault in shard 1, shards 0 and 2 are byte-equal to an undisturbed run and to the oracle, shard 1 restarted is
byte-equal to fresh streams."""
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


@pytest.fixture(scope="module")
def FI(M):
    if not M.FAULT_LIB_PATH.exists():              # (a checkout without built artefacts: the test build is one `make fault` away -- host files only)
        M.build()
    return M.load_fault_library()


def _cfgs(M, streams):
    return [M.StreamConfig(samplerate=r, mode=m, bitrate=k, psy_model=p) for r, m, k, p in streams]


def _oracle(pcm_s, c):
    return O.oracle_stream(pcm_s, samplerate=c.samplerate, mode=c.mode, kbps=c.bitrate, psy=c.psy_model)[0]


def _planar(inter_s, c):
    T = inter_s.shape[0]
    if c.mode == "m":
        return np.repeat(inter_s[:, None, :1152], 2, axis=1)
    return inter_s.reshape(T, 1152, 2).transpose(0, 2, 1)


def test_product_library_has_no_fault_hooks(M):
    """the hooks exist in the TEST build only"""
    L = M.load_library()
    for name in ("tlb_debug_fail_next", "tlb_debug_tick_fail_next", "tlb_debug_node_fail_next"):
        assert not hasattr(L, name), name


def test_batch_broken_by_a_failed_launch_refuses_until_reset(M, FI):
    """tlb_batch: an injected half-way failure marks the batch broken (every launch refused), tlb_reset() -- which now rebuilds the
    device's stream tables, lists and pairing from the host's first -- brings it back as a fresh batch: oracle bytes."""
    cfgs = _cfgs(M, MIX[:6])
    pcm = np.stack([gen_pcm(8100 + s, (0, 7, 5, 4)[s % 4], 0, 4) for s in range(len(cfgs))], axis=1)
    b = M.Batch(cfgs, lib=FI)
    b.encode(pcm[:1])
    assert b.fail_next(1) == 0
    with pytest.raises(M.ToolameError) as e:
        b.encode(pcm[1:2])
    assert e.value.code == 17
    with pytest.raises(M.ToolameError):                                 # sticky
        b.encode(pcm[1:2])
    b.reset()
    got, _ = b.encode(pcm)
    tail = b.flush()
    b.close()
    for s, c in enumerate(cfgs):
        assert got[s] + tail[s] == _oracle(pcm[:, s], c), s


def test_tick_failure_is_sticky(M, FI):
    """a submit that fails in its LAST group leaves the earlier groups one frame ahead; the object says so and refuses
    every further call instead of emitting a duplicate frame on a retry."""
    cfgs = _cfgs(M, [(48000, "s", 128, 1)] * 6)
    inter = np.stack([np.stack([gen_pcm(8200 + s, 0, 0, 3)[f].T.reshape(-1) for s in range(6)]) for f in range(3)])
    t = M.Tick(cfgs, egress="frames", ngroups=3, lib=FI)
    t.pcm[:] = inter[0]
    t.run()
    assert t.status() == 0
    first = [t.frame(s) for s in range(6)]
    t.fail_next(1)
    t.pcm[:] = inter[1]
    with pytest.raises(M.ToolameError) as e:
        t.run()
    assert e.value.code == 17 and t.status() == 17
    assert t.pcm is None                                                # no input set is handed out any more
    for call in (t.run, t.submit, t.finish, lambda: t.stream_reset(0)):
        with pytest.raises(M.ToolameError) as e2:
            call()
        assert e2.value.code == 17
    assert [t.frame(s) for s in range(6)] == first                      # the read accessors keep showing the last tick waited for
    t.close()


@pytest.mark.parametrize("when", [2, 4])
def test_batch_plane_fault_in_shard_1_leaves_shards_0_and_2_untouched(M, FI, when):
    """devices = (0, 0, 0), 13 mixed streams, 7 calls of one frame + flush.  Shard 1's launch number `when` fails: the call reports it,
    shards 0 and 2 finish THAT call and every later one byte-equal to an undisturbed node and to the oracle; shard 1 answers nothing
    while broken; restarted it encodes the remaining input as fresh streams (oracle from that frame on)."""
    cfgs = _cfgs(M, MIX)
    ns, nf = len(cfgs), 7
    pcm = np.stack([gen_pcm(8300 + s, (0, 7, 5, 4)[s % 4], 0, nf) for s in range(ns)], axis=1)
    ref = M.Node(cfgs, devices=(0, 0, 0), plane="batch")
    want = [ref.encode(pcm[f:f + 1]) for f in range(nf)]
    want_tail = ref.flush()
    ref.close()
    nd = M.Node(cfgs, devices=(0, 0, 0), plane="batch", lib=FI)
    print(nd.describe())
    blocks = nd.blocks
    (f1, n1) = blocks[1]
    in1 = lambda s: f1 <= s < f1 + n1
    nd.fail_next(1, when)
    got, restarted_at = [], None
    for f in range(nf):
        if f + 1 == when:
            with pytest.raises(M.ToolameError) as e:
                nd.encode(pcm[f:f + 1])
            assert e.value.code == 17
            st = nd.shard_status(1)
            assert not st["ok"] and st["state"] == 1 and st["last_err"] == 17 and st["failures"] == 1 and "tlb_encode_device_len" in st["what"]
            assert nd.shard_status(0)["ok"] and nd.shard_status(2)["ok"]
            nd.sync()                                                   # the healthy shards' launches of the failing call
            got.append(nd.download())
            # a stream of the broken shard: life-cycle calls are refused, nothing is read
            with pytest.raises(M.ToolameError):
                nd.stream_reset(f1)
        elif f + 1 == when + 1:
            got.append(nd.encode(pcm[f:f + 1]))                         # one more step with the shard down: returns 0, shard 1 silent
            nd.shard_restart(1)
            restarted_at = f + 1
            st = nd.shard_status(1)
            assert st["ok"] and st["restarts"] == 1 and st["last_err"] == 17
        else:
            got.append(nd.encode(pcm[f:f + 1]))
    tail = nd.flush()
    per, tot = nd.counters()
    nd.close()
    for s in range(ns):
        if not in1(s):
            for f in range(nf):
                assert got[f][s] == want[f][s], (f, s)
            assert tail[s] == want_tail[s], s
            assert b"".join(g[s] for g in got) + tail[s] == _oracle(pcm[:, s], cfgs[s]), s
        else:
            for f in range(when - 1):
                assert got[f][s] == want[f][s], (f, s)                  # before the fault: as everyone
            assert got[when - 1][s] == b"" and got[when][s] == b"", s   # down: silent
            fresh = _oracle(pcm[restarted_at:, s], cfgs[s])             # after the restart: a freshly started encoder on the remaining input
            assert b"".join(g[s] for g in got[restarted_at:]) + tail[s] == fresh, s
    # counters stay those of completed steps: shards 0 and 2 all seven, shard 1 those before the fault and after the restart
    assert per[0]["steps"] == nf and per[2]["steps"] == nf and per[1]["steps"] == (when - 1) + (nf - restarted_at)
    assert per[1]["frames"] == n1 * per[1]["steps"] and tot["frames"] == sum(p["frames"] for p in per)


def test_tick_plane_fault_in_shard_1_restart_joins_the_lockstep(M, FI):
    """The real-time loop: three shards on the one GPU, ticks overlapped; shard 1's third tick fails.  Shards 0 and 2 deliver every tick
    exactly as an undisturbed node (frames, levels, silence counters); shard 1 is silent until restarted, then delivers the frames of a
    freshly started encoder, in step with the others."""
    streams = [(48000, "s", 128, 1), (48000, "j", 128, 3), (24000, "m", 64, 1), (48000, "s", 192, 2), (48000, "m", 96, 4), (48000, "m", 96, 4),
               (16000, "m", 32, 3), (48000, "s", 128, 1), (48000, "j", 160, 3)]
    cfgs = _cfgs(M, streams)
    ns, T = len(cfgs), 9
    inter = np.stack([np.stack([gen_pcm(8400 + s, (0, 7, 5, 4)[s % 4], 0, T)[f].T.reshape(-1) for s in range(ns)]) for f in range(T)])
    snap = lambda t: [(t.frame(s), t.peaks(s), t.silence_ms(s)) for s in range(ns)]
    ref = M.Node(cfgs, devices=(0, 0, 0), plane="tick", egress="frames", ngroups=2)
    want = []
    for f in range(T):
        ref.set_pcm(inter[f])
        ref.run()
        want.append(snap(ref))
    ref.finish()
    want.append(snap(ref))
    ref.close()
    nd = M.Node(cfgs, devices=(0, 0, 0), plane="tick", egress="frames", ngroups=2, lib=FI)
    (f1, n1) = nd.blocks[1]
    in1 = lambda s: f1 <= s < f1 + n1
    nd.fail_next(1, 3)
    got, restarted_at = [], None
    for f in range(T):
        nd.set_pcm(inter[f])
        if f == 2:
            with pytest.raises(M.ToolameError) as e:
                nd.run()
            assert e.value.code == 17 and not nd.shard_ok(1) and nd.shard_ok(0) and nd.shard_ok(2)
            assert "tlb_tick_submit" in nd.shard_status(1)["what"]
            assert nd.pcm(f1) is None and nd.peaks(f1) is None and nd.frame(f1) == b""
        elif f == 4:
            nd.shard_restart(1, now_s=1712345999)
            restarted_at = f
            nd.set_pcm(inter[f])                                        # the restarted shard has a fresh input set
            nd.run()
        else:
            nd.run()
        got.append(snap(nd))
    nd.finish()
    got.append(snap(nd))
    per, tot = nd.counters()
    nd.close()
    for s in range(ns):
        if not in1(s):
            for f in range(T + 1):
                assert got[f][s] == want[f][s], (f, s)
        else:
            for f in range(2):
                assert got[f][s] == want[f][s], (f, s)
            for f in (2, 3):
                assert got[f][s] == (b"", None, 0), (f, s)
            fresh = _oracle(_planar(inter[restarted_at:, s], cfgs[s]), cfgs[s])
            assert b"".join(got[f][s][0] for f in range(restarted_at, T + 1)) == fresh, s
    assert per[0]["steps"] == T and per[2]["steps"] == T and per[1]["steps"] == 2 + (T - restarted_at)


def test_every_shard_broken_is_reported_and_restartable(M, FI):
    cfgs = _cfgs(M, [(48000, "s", 128, 1)] * 4)
    pcm = np.stack([gen_pcm(8500 + s, 0, 0, 2) for s in range(4)], axis=1)
    nd = M.Node(cfgs, devices=(0, 0), plane="batch", lib=FI)
    nd.fail_next(0, 1)
    nd.fail_next(1, 1)
    with pytest.raises(M.ToolameError):
        nd.encode(pcm[:1])
    assert not nd.shard_ok(0) and not nd.shard_ok(1)
    nd.upload(pcm[:1])
    with pytest.raises(M.ToolameError) as e:                            # nobody left: the node says so instead of returning 0
        nd.encode_resident()
    assert e.value.code == 17
    assert nd.L.tlb_node_sync(nd.h) == 17                               # retires the node's step clock; no live shard: TLB_ERR_HIP again
    nd.shard_restart(0)
    nd.shard_restart(1)
    got = nd.encode(pcm)
    tail = nd.flush()
    nd.close()
    for s in range(4):
        assert got[s] + tail[s] == _oracle(pcm[:, s], cfgs[s]), s
