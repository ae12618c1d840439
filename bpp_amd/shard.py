"""Sharding of loci over ranks (one process per GPU) and the one collective the path needs.

The reference shards loci over pthreads in contiguous ranges, optionally after a zig-zag
sort by work (threads.c:234-353) and reduces 1-4 scalars per all-loci proposal
(threads.c:525-591).  Here: the same partition over ranks, and a sum all-reduce of the packed
per-proposal scalars through torch.distributed (backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests).
"""
import numpy as np


def partition(work, nranks, zigzag=True):
    """work[i] = cost of locus i (tips * patterns).  Returns a list of index arrays, one per
    rank.  zigzag: sort by work descending and deal 0..n-1, n-1..0, ... (load_balance_zigzag,
    threads.c:265-353); else contiguous equal-count ranges (load_balance_none, threads.c:234)."""
    work = np.asarray(work)
    n = len(work)
    if nranks <= 0:
        raise ValueError("nranks must be positive")
    if not zigzag:
        bounds = [(n * r) // nranks for r in range(nranks + 1)]
        return [np.arange(bounds[r], bounds[r + 1]) for r in range(nranks)]
    order = np.argsort(-work, kind="stable")
    out = [[] for _ in range(nranks)]
    for j, idx in enumerate(order):
        rnd, pos = divmod(j, nranks)
        out[pos if rnd % 2 == 0 else nranks - 1 - pos].append(int(idx))
    return [np.array(sorted(o), dtype=np.int64) for o in out]


def allreduce_sum(values, dist=None, device_tensor=None):
    """sum of a small vector over all ranks; values: list/array of doubles (ignored when a
    device tensor that already holds them is given)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(values, dtype=np.float64) if device_tensor is None else device_tensor
    import torch
    if device_tensor is not None:
        dist.all_reduce(device_tensor)
        return device_tensor
    t = torch.tensor(np.asarray(values, dtype=np.float64))
    dist.all_reduce(t)
    return t.numpy()
