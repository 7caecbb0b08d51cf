"""Tile sharding across GPUs: tiles are independent, so tile i of a
batch goes to rank i mod G and nothing but a tile count crosses xGMI."""
import numpy as np


def shard_indices(n_tiles, rank, world):
    """Indices of the global batch rendered by `rank`: i with i mod world == rank."""
    return np.arange(rank, n_tiles, world)


def reduce_tile_count(local_count, dist=None, device=None):
    """Sum of per-rank tile counts: one all-reduce of a single int64 (RCCL when the
    process group is 'nccl', gloo on CPU).  Returns the global count."""
    import torch

    t = torch.tensor([int(local_count)], dtype=torch.int64, device=device)
    if dist is not None and dist.is_initialized():
        dist.all_reduce(t)
    return int(t.item())
