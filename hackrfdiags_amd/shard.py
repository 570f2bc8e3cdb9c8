"""Multi-GPU plumbing: channels are independent, so N GPUs = N
contiguous channel shards, one process per GPU, per-channel state pinned to its
GPU for the life of the stream.  There is no collective on the data path when the
IQ is fed per GPU; when all IQ lands on rank 0 it is scattered once per batch
(RCCL over xGMI: a one-to-all scatter uses the 7 point-to-point links out of rank 0
concurrently) and the PCM (1 KiB per channel-block) is gathered back.

Works with backend "nccl" (= RCCL on ROCm, GPU tensors) and "gloo" (CPU tensors;
used by the CPU tests of this plumbing)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def channel_range(rank: int, world: int, n_channels: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank; the first n_channels % world ranks get one more."""
    base, extra = divmod(n_channels, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def scatter_iq(iq_all: Optional[torch.Tensor], n_channels: int, blocks: int, block_bytes: int,
               device: torch.device, src: int = 0) -> torch.Tensor:
    """Rank `src` holds int8 [n_channels, blocks, block_bytes]; every rank gets its shard."""
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = channel_range(rank, world, n_channels)
    mine = torch.empty((hi - lo, blocks, block_bytes), dtype=torch.int8, device=device)
    if world == 1:
        mine.copy_(iq_all)
        return mine
    # shards may differ by one channel: point-to-point sends, all links busy at once
    if rank == src:
        reqs = []
        for r in range(world):
            a, b = channel_range(r, world, n_channels)
            if r == src:
                mine.copy_(iq_all[a:b])
            elif b > a:
                reqs.append(dist.isend(iq_all[a:b].contiguous(), dst=r))
        for q in reqs:
            q.wait()
    elif hi > lo:
        dist.recv(mine, src=src)
    return mine


def gather_pcm(pcm_mine: torch.Tensor, n_channels: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Inverse of scatter_iq for the int16 [shard, blocks, n_pcm] output; rank dst gets all."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if world == 1:
        return pcm_mine
    out = None
    if rank == dst:
        out = torch.empty((n_channels,) + tuple(pcm_mine.shape[1:]), dtype=pcm_mine.dtype, device=pcm_mine.device)
        reqs = []
        for r in range(world):
            a, b = channel_range(r, world, n_channels)
            if r == dst:
                out[a:b].copy_(pcm_mine)
            elif b > a:
                reqs.append((dist.irecv(out[a:b], src=r)))
        for q in reqs:
            q.wait()
    elif pcm_mine.shape[0] > 0:
        dist.send(pcm_mine.contiguous(), dst=dst)
    return out


def max_over_ranks(seconds: float, device: torch.device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
