"""Multi-GPU sharding of the generator-driven loop: one process per GPU, scene replicated.

Cells of one frame are dealt round-robin over ranks (`cell = rank + m * world`), which balances load without
any knowledge of the scene; the only exchange is one all-gather of equal-sized shards per frame.  The
fixed-point accumulation makes every cell's value independent of who computed it, so the assembled image is
bit-identical to the single-GPU image.
"""
import numpy as np


def cell_shard(rank, world, n_cells):
    """(cell_begin, cell_count, cell_step) of `rank`: cells rank, rank+world, ..."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    count = (n_cells - rank + world - 1) // world if n_cells > rank else 0
    return rank, count, world


def shard_capacity(world, n_cells):
    """Rows every rank allocates so the all-gather has equal-sized inputs."""
    return (n_cells + world - 1) // world


def interleave_shards(shards, n_cells):
    """[world, capacity, ...] gathered shards -> [n_cells, ...] in cell order (torch tensor or numpy array)."""
    world, cap = shards.shape[0], shards.shape[1]
    if hasattr(shards, "transpose") and not isinstance(shards, np.ndarray):  # torch
        full = shards.transpose(0, 1).reshape((cap * world,) + tuple(shards.shape[2:]))
    else:
        full = np.swapaxes(shards, 0, 1).reshape((cap * world,) + tuple(shards.shape[2:]))
    return full[:n_cells]


def gather_image(local, rank, world, n_cells, dist, dst=None):
    """All-gather equal-sized shards (torch tensors, padded to shard_capacity) and return the full image.

    `dist` is torch.distributed (backend nccl = RCCL on the GPU box, gloo in the CPU tests)."""
    import torch

    cap = shard_capacity(world, n_cells)
    if local.shape[0] != cap:
        pad = torch.zeros((cap - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    out = torch.empty((world * cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return interleave_shards(out.reshape((world, cap) + tuple(local.shape[1:])), n_cells)
