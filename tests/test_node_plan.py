"""Node level (include/toolame_batch.h part 3, csrc/tlb_node.cpp) without a GPU: the partition of SURVEY section 8e -- contiguous
stream blocks [g*N/G, (g+1)*N/G) -- and what each block turns into inside its device (distinct configurations, one kernel list
per psy model, mono streams in pairs).  Pure host arithmetic through the C-ABI; the GPU side is tests/test_node_gpu.py."""
import ctypes as C

import numpy as np
import pytest

import odr_audioenc_amd as M


@pytest.mark.parametrize("n,g", [(131072, 8), (16384, 8), (10, 3), (7, 7), (1000003, 8), (5, 2), (4097, 4), (1, 1)])
def test_partition_is_contiguous_and_balanced(n, g):
    blocks = M.node_partition(n, g)
    assert len(blocks) == g
    pos = 0
    for k, (first, cnt) in enumerate(blocks):
        assert first == pos == n * k // g                      # [g*N/G, (g+1)*N/G)
        pos += cnt
    assert pos == n
    sizes = [c for _, c in blocks]
    assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 1


def test_partition_rejects_nonsense():
    L = M.load_library()
    f, c = C.c_int(-1), C.c_int(-1)
    for args in [(0, 4, 0), (10, 0, 0), (10, 4, 4), (10, 4, -1)]:
        L.tlb_node_partition(*args, C.byref(f), C.byref(c))
        assert (f.value, c.value) == (0, 0), args


def test_plan_configs4_grouping_inside_each_device():
    """BASELINE configs[4]: 32 kHz mono 64 kbps and 48 kHz stereo 192 kbps interleaved, psy 4, over 8 shards with N not divisible
    by 8: every shard holds both configurations (2 records), ONE homogeneous kernel list (psy 4 rides on list 2) whose size is the
    block's, and its mono streams pair up (the odd one out stays alone)."""
    n, g = 16387, 8
    cfgs = [M.StreamConfig(samplerate=32000, mode="m", bitrate=64, psy_model=4) if s % 2 == 0 else
            M.StreamConfig(samplerate=48000, mode="s", bitrate=192, psy_model=4) for s in range(n)]
    plan = M.node_plan(cfgs, g)
    assert [(p["first"], p["n"]) for p in plan] == M.node_partition(n, g)
    for p in plan:
        assert p["nconfigs"] == 2 and p["lists"] == [0, 0, p["n"], 0]
        monos = sum(1 for s in range(p["first"], p["first"] + p["n"]) if s % 2 == 0)
        assert p["mono_pairs"] == monos // 2
    assert sum(p["n"] for p in plan) == n


def test_plan_mixed_models_and_rates():
    rng = np.random.default_rng(5)
    choices = [(48000, "s", 128, 1), (48000, "j", 128, 3), (24000, "m", 64, 1), (48000, "s", 192, 2), (48000, "m", 96, 4), (16000, "m", 32, 3),
               (48000, "m", 64, 0), (48000, "d", 128, 0)]
    pick = rng.integers(0, len(choices), 1001)
    cfgs = [M.StreamConfig(samplerate=choices[i][0], mode=choices[i][1], bitrate=choices[i][2], psy_model=choices[i][3]) for i in pick]
    for g in (1, 2, 3, 8):
        plan = M.node_plan(cfgs, g)
        tot = np.zeros(4, dtype=int)
        for p in plan:
            blk = pick[p["first"]:p["first"] + p["n"]]
            assert p["nconfigs"] == len(set(blk.tolist()))
            want = [0, 0, 0, 0]
            for i in blk:
                m = choices[i][3]
                want[2 if m == 4 else m] += 1
            assert p["lists"] == want
            pairs = sum(int(np.sum(blk == i)) // 2 for i in range(len(choices)) if choices[i][1] == "m")
            assert p["mono_pairs"] == pairs
            tot += np.array(p["lists"])
        assert tot.sum() == len(cfgs)


def test_plan_reports_the_first_illegal_configuration():
    cfgs = [M.StreamConfig() for _ in range(10)]
    cfgs[7] = M.StreamConfig(bitrate=100)                       # not an MPEG-1 Layer II rate (common.c:95-116)
    plan_ok = M.node_plan(cfgs[:7], 2)
    assert len(plan_ok) == 2
    with pytest.raises(M.ToolameError) as e:
        M.node_plan(cfgs, 2)
    assert e.value.code == 4                                    # TLB_ERR_BITRATE
    cfgs[7] = M.StreamConfig(samplerate=48000, mode="m", bitrate=32, pad_len=200)    # PAD larger than the frame leaves room for
    with pytest.raises(M.ToolameError) as e:
        M.node_plan(cfgs, 3)
    assert e.value.code == 5                                    # TLB_ERR_PAD


def test_node_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(M.ToolameError) as e:
        M.Node([M.StreamConfig() for _ in range(4)], devices=(0, 0))
    assert e.value.code == 16                                   # TLB_ERR_NO_DEVICE: no CPU fallback
