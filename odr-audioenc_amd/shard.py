"""Multi-GPU sharding of the encode path: streams are independent, so the unit of
sharding is the stream and there is NO data-path collective.  One process per GPU; torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests) carries only the barrier
and two scalars per measurement window (max elapsed time, total frames)."""
import os
import time


def init_from_env(backend=None, device_index=None, force=False):
    """-> (rank, local_rank, world, dist or None).  Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*.
    force: build the process group at world size 1 too -- a one-rank group is still a group: init_process_group, barrier, all_reduce and
    all_gather go through the backend's code (RCCL for "nccl"), which is how the collective path gets executed on a one-GPU box."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and not force:
        return rank, local_rank, world, None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:          # only a forced one-rank group gets here without a launcher's port
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        s.close()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank if device_index is None else device_index)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world, dist


def strong_range(rank, world, total_streams):
    """contiguous block [lo, hi) of `total_streams` owned by `rank` (strong scaling split)."""
    return rank * total_streams // world, (rank + 1) * total_streams // world


def weak_stream_ids(rank, streams_per_rank):
    """global stream ids (= PCM seeds) of a rank when every rank owns `streams_per_rank` streams."""
    return range(rank * streams_per_rank, (rank + 1) * streams_per_rank)


def barrier(dist, device_sync=None):
    if device_sync:
        device_sync()
    if dist is not None:
        dist.barrier()
    if device_sync:
        device_sync()


def timed_region(dist, fn, device_sync=None, device="cpu"):
    """barrier + sync, run fn(), barrier + sync; returns MAX over ranks of the elapsed seconds."""
    return timed_region_detail(dist, fn, device_sync, device)[0]


def timed_region_detail(dist, fn, device_sync=None, device="cpu"):
    """As timed_region; returns (MAX over ranks of the barrier-to-barrier seconds, this rank's own seconds from the
    opening barrier to the end of its own device work -- what the per-GPU rate is computed from)."""
    barrier(dist, device_sync)
    t0 = time.perf_counter()
    fn()
    if device_sync:
        device_sync()
    own = time.perf_counter() - t0
    barrier(dist, device_sync)
    elapsed = time.perf_counter() - t0
    return reduce_max(dist, elapsed, device), own


def gather_floats(dist, values, device="cpu"):
    """all_gather of a short list of floats per rank -> [world][len(values)] (a few bytes: the only traffic besides
    the barriers; on the GPU box it travels over RCCL/xGMI)."""
    if dist is None:
        return [[float(v) for v in values]]
    import torch
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(x) for x in o.cpu().tolist()] for o in out]


def gather_bytes(dist, payload, width=160, device="cpu"):
    """all_gather of one short byte string per rank (padded / cut to `width`) -> [world] byte strings, over the SAME backend as the
    barriers: how a record of a multi-GPU run shows which device every rank really ran on (a GPU's UUID and PCI address), gathered by
    the collective itself rather than inferred from the launcher's environment."""
    payload = bytes(payload)[:width]
    if dist is None:
        return [payload]
    import torch
    t = torch.zeros(width, dtype=torch.uint8, device=device)
    if payload:
        t[:len(payload)] = torch.tensor(list(payload), dtype=torch.uint8, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [bytes(o.cpu().tolist()).rstrip(b"\0") for o in out]


def reduce_max(dist, value, device="cpu"):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum(dist, value, device="cpu"):
    if dist is None:
        return int(value)
    import torch
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
