"""Multi-process helpers for the frame-sharded path (one process per GPU).

IDR pictures are independent units, so ranks never exchange picture data: the only
cross-rank traffic is control (barrier, max-over-ranks of the clock, gathering
per-rank checksums).  backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests."""
import os


def shard(n_items, rank, world):
    """Contiguous, balanced slice [lo, hi) of n_items owned by `rank`."""
    lo = n_items * rank // world
    hi = n_items * (rank + 1) // world
    return lo, hi


def init(backend, rank=None, world=None, device_id=None):
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    kw = {}
    if device_id is not None:
        kw["device_id"] = device_id
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def max_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_objects(obj):
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out
