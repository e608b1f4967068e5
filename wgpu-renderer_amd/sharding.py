"""Splitting frames across ranks (one process per GPU).

Two ways, both without any collective inside the render path:
  * frame-parallel: the poses of a batch are split contiguously over the ranks;
  * screen-bin sharding: the 64x64-pixel bins of one frame are dealt round-robin
    (bin b belongs to rank b % world, WrConfig.tile_rank / tile_world), each rank renders
    only its bins, and the packed colour bins are exchanged once with all_gather.

The numpy pack / scatter functions restate the device kernels behind wr_pack_owned_bins and
wr_scatter_bins so that the exchange logic can be tested on CPU with the gloo backend.
"""
from __future__ import annotations

from typing import List

import numpy as np

BIN = 64


def pose_shard(n_poses: int, rank: int, world: int) -> range:
    """Contiguous shard of a pose batch; the first n_poses % world ranks get one more."""
    base, extra = divmod(n_poses, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def bin_grid(width: int, height: int):
    return (width + BIN - 1) // BIN, (height + BIN - 1) // BIN


def owned_bins(width: int, height: int, rank: int, world: int) -> List[int]:
    bx, by = bin_grid(width, height)
    return list(range(rank, bx * by, world))


def bins_per_rank(width: int, height: int, world: int) -> int:
    """Packed-buffer capacity in bins, equal on every rank (all_gather needs equal sizes)."""
    bx, by = bin_grid(width, height)
    return (bx * by + world - 1) // world


def pack_bins_host(frame: np.ndarray, rank: int, world: int) -> np.ndarray:
    """frame: (H, W) uint32 -> (bins_per_rank, 64, 64) uint32, zero padded."""
    h, w = frame.shape
    bx, _ = bin_grid(w, h)
    out = np.zeros((bins_per_rank(w, h, world), BIN, BIN), np.uint32)
    for j, b in enumerate(owned_bins(w, h, rank, world)):
        oy, ox = (b // bx) * BIN, (b % bx) * BIN
        tile = frame[oy:oy + BIN, ox:ox + BIN]
        out[j, :tile.shape[0], :tile.shape[1]] = tile
    return out


def scatter_bins_host(frame: np.ndarray, packed: np.ndarray, rank: int, world: int) -> None:
    h, w = frame.shape
    bx, _ = bin_grid(w, h)
    for j, b in enumerate(owned_bins(w, h, rank, world)):
        oy, ox = (b // bx) * BIN, (b % bx) * BIN
        th, tw = min(BIN, h - oy), min(BIN, w - ox)
        frame[oy:oy + th, ox:ox + tw] = packed[j, :th, :tw]


def gather_packed(packed, world: int):
    """all_gather of one rank's packed bins (torch tensor, CPU/gloo or GPU/RCCL).  Returns a
    tensor of shape (world, *packed.shape)."""
    import torch
    import torch.distributed as dist

    out = torch.empty((world,) + tuple(packed.shape), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out.view(-1), packed.reshape(-1).contiguous())
    return out


def max_over_ranks(seconds: float, device=None) -> float:
    """The slowest rank's time, as bench.py reports it."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
