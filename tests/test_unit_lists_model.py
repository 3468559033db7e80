"""The arithmetic of the per-XCD unit lists (csrc/toolame_hip.hip, tl_take_unit) restated in Python and checked for what it must
guarantee whatever the launch shape and whatever order the waves arrive in: every (stream, frame) unit of a launch is handed out
exactly once.  (The device function itself is exercised by the -m gpu parity tests, test_unit_lists_odd_shapes among them.)"""
import random

import pytest

BLOCK_LOG2 = 5


def run_launch(nlist, nframes, grid, waves_per_block, rng):
    nunits = nlist * nframes
    nblocks = (nunits + (1 << BLOCK_LOG2) - 1) >> BLOCK_LOG2
    heads = [0] * 8
    lim = [((nblocks - q + 7) >> 3) << BLOCK_LOG2 for q in range(8)]
    ng = [((grid - q + 7) >> 3) * waves_per_block if grid > q else 0 for q in range(8)]
    taken = []
    waves = [{"grp": b & 7, "hop": 0, "first": (b >> 3) * waves_per_block + w, "done": False} for b in range(grid) for w in range(waves_per_block)]

    def take(wv):
        while wv["hop"] < 8:
            q = (wv["grp"] + wv["hop"]) & 7
            if wv["first"] >= 0:
                v, wv["first"] = wv["first"], -1
            else:
                if wv["hop"] > 0 and ng[q] + heads[q] >= lim[q]:
                    wv["hop"] += 1
                    continue
                v = ng[q] + heads[q]
                heads[q] += 1
            if v >= lim[q]:
                wv["hop"] += 1
                continue
            u = ((((v >> BLOCK_LOG2) << 3) + q) << BLOCK_LOG2) + (v & ((1 << BLOCK_LOG2) - 1))
            if u >= nunits:
                continue
            return divmod(u, nframes)
        return None

    live = list(waves)
    while live:
        wv = live[rng.randrange(len(live))]            # any interleaving of the waves
        r = take(wv)
        if r is None:
            live.remove(wv)
        else:
            taken.append(r)
    return taken


@pytest.mark.parametrize("nlist,nframes", [(1, 1), (1, 5), (1, 8192 // 64), (3, 7), (7, 1), (8, 32), (9, 2), (37, 5), (100, 12), (4096 // 16, 8), (513, 3)])
def test_every_unit_exactly_once(nlist, nframes):
    rng = random.Random(nlist * 1000 + nframes)
    for grid, wpb in ((1, 12), (3, 12), (8, 12), (21, 12), (64, 4)):
        got = run_launch(nlist, nframes, grid, wpb, rng)
        assert len(got) == nlist * nframes and len(set(got)) == len(got), (grid, wpb)
        assert set(got) == {(k, f) for k in range(nlist) for f in range(nframes)}


# ---- the psy-2 kernel's work list (csrc/mp2_host.cpp tl_psy2_plan, csrc/mp2_wave.h tl_psy2_unit): whole chains, then runs of frames ----
@pytest.mark.parametrize("slots", [1, 3, 12, 3072])
def test_psy2_runs_cover_every_frame_of_every_chain_once(slots):
    import ctypes as C
    import numpy as np
    import emulib
    L = emulib.lib()
    L.emu_psy2_units.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    rng = random.Random(slots)
    shapes = [(1, 1), (2, 1), (2, 16), (2, 8192), (5, 7), (8192, 32), (8192, 1), (3073, 8), (4096, 3), (24576, 8), (6144, 32), (7, 33)]
    shapes += [(rng.randrange(1, 9000), rng.randrange(1, 70)) for _ in range(40)]
    for nchain, nframes in shapes:
        cap = nchain * max(1, nframes)
        units = np.zeros((cap, 3), dtype=np.int32)
        n = L.emu_psy2_units(nchain, nframes, slots, units.ctypes.data, cap)
        assert 0 < n <= cap, (nchain, nframes, slots, n)
        u = units[:n]
        u = u[u[:, 2] > u[:, 1]]                               # empty runs are skipped by the kernel
        assert (u[:, 1] >= 0).all() and (u[:, 2] <= nframes).all() and (u[:, 0] >= 0).all() and (u[:, 0] < nchain).all()
        cover = np.zeros((nchain, nframes + 1), dtype=np.int32)
        np.add.at(cover, (u[:, 0], u[:, 1]), 1)
        np.add.at(cover, (u[:, 0], u[:, 2]), -1)
        assert (np.cumsum(cover, axis=1)[:, :nframes] == 1).all(), (nchain, nframes, slots)
        # longest first: whole chains come before every run, and no plan is slower than one unit per chain
        lens = (units[:n, 2] - units[:n, 1])
        whole = lens == nframes
        assert not whole[np.argmin(whole):].any() or whole.all()
        rounds = lambda k: -(-k // slots)
        t_plain = rounds(nchain) * nframes
        nw = int(whole.sum()) if not whole.all() else nchain
        t_plan = rounds(nw) * nframes if whole.all() else (nw // slots) * nframes + rounds(n - nw) * (lens[~whole].max() + 0.5)
        assert t_plan <= t_plain + 1e-9, (nchain, nframes, slots, t_plan, t_plain)
