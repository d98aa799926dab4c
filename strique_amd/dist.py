"""Multi-GPU plumbing: reads shard across ranks, results are gathered once at the end.

Every (read, target) is independent (reference scripts/STRique.py:702-704; the reference itself
spreads reads over worker processes, :743-746), so there is no collective on the data path.  The
only exchange is the final gather of fixed-size result records to rank 0 -- with the "nccl" backend
that is RCCL over xGMI (~64 B per read).
"""
import os

import numpy as np


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_indices(n_items, rank, world, cost=None):
    """Indices handled by `rank`: items sorted by descending cost (read length) dealt round-robin,
    which balances the sequential DP work per GPU.  Deterministic on every rank."""
    order = np.arange(n_items) if cost is None else np.argsort(-np.asarray(cost), kind="stable")
    return np.sort(order[rank::world])


def init_process_group(backend=None):
    import torch
    import torch.distributed as dist
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def gather_records(records, index, n_total, device=None):
    """records: structured numpy array of this rank's results, index: their global positions.
    Returns the full array on rank 0 (None elsewhere).  One all_gather of (index, payload)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        out = np.zeros(n_total, dtype=records.dtype)
        out[index] = records
        return out
    world = dist.get_world_size()
    itemsize = records.dtype.itemsize
    counts = [None] * world
    dist.all_gather_object(counts, int(len(records)))
    cap = max(counts)
    payload = np.zeros((cap, itemsize), np.uint8)
    payload[:len(records)] = records.view(np.uint8).reshape(len(records), itemsize)
    idx = np.full(cap, -1, np.int64); idx[:len(records)] = index
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    t_pay = torch.from_numpy(payload).to(dev); t_idx = torch.from_numpy(idx).to(dev)
    pays = [torch.empty_like(t_pay) for _ in range(world)]; idxs = [torch.empty_like(t_idx) for _ in range(world)]
    dist.all_gather(pays, t_pay)
    dist.all_gather(idxs, t_idx)
    if dist.get_rank() != 0:
        return None
    out = np.zeros(n_total, dtype=records.dtype)
    for r in range(world):
        k = counts[r]
        ii = idxs[r][:k].cpu().numpy()
        out[ii] = pays[r][:k].cpu().numpy().reshape(-1).view(records.dtype)
    return out
