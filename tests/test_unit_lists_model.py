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
