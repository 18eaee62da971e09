"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" = RCCL on ROCm).

The factor/solve path shards by INDEPENDENT systems / independent elimination-tree subtrees: there is
no data-path collective in this round (multi-GPU), only the timing reduction bench.py
needs.  `gloo` works for the same code on CPU (tests/test_dist_cpu.py, world_size 2).
"""
import os


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment; returns (rank, world, dist or None)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world <= 1:
        return 0, 1, None
    import torch
    import torch.distributed as dist
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend, **kw)
    return rank, world, dist


def shard(n_items, rank, world):
    """Contiguous block of `n_items` independent work units owned by `rank` (first ranks get the remainder)."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(value, dist, device="cpu"):
    """Max of a python float over all ranks (the timing rule of bench.py)."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist, device="cpu"):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
