"""Multi-GPU plumbing for the batch-sharded inference path (one process per GPU).

Utterances are independent, so a global batch is split into contiguous
per-rank shards and there is NO data-path collective; the only cross-rank traffic is control
plane (barrier, max of elapsed time, gathering results on rank 0), carried by torch.distributed
(gloo on CPU tensors -- the engine's device buffers never go through torch).
"""
import os

import numpy as np


def env_rank():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend="gloo"):
    rank, local_rank, world = env_rank()
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if not dist.is_initialized():
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_bounds(n, rank, world):
    """Contiguous, balanced [lo, hi) of n utterances for `rank` (first n % world ranks get one more)."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_batch(batch, rank, world):
    """Slice every per-utterance array of a batch dict (leading dim = utterance)."""
    n = len(batch["text_lengths"])
    lo, hi = shard_bounds(n, rank, world)
    return {k: (v[lo:hi] if isinstance(v, np.ndarray) and v.shape[:1] == (n,) else v) for k, v in batch.items()}


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(x):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_to_rank0(array):
    """Concatenate per-rank host arrays (equal trailing dims) on rank 0; None elsewhere."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return array
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(array, out, dst=0)
    return np.concatenate(out, 0) if out is not None else None


def broadcast_bytes(data, src=0):
    """Control-plane broadcast of a short byte string (the 128-byte RCCL unique id of data-parallel training)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return data
    box = [data if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def mean_over_ranks(x):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item()) / dist.get_world_size()
