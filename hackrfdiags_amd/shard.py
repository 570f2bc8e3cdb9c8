"""Multi-GPU plumbing: channels are independent, so N GPUs = N
contiguous channel shards, one process per GPU, per-channel state pinned to its
GPU for the life of the stream.  There is no collective on the data path when the
IQ is fed per GPU.  When all IQ lands on rank 0 (the north star's "per-channel
scatter") it leaves rank 0 as ONE group of point-to-point sends
(`dist.batch_isend_irecv` = ncclGroupStart ... ncclGroupEnd under RCCL): the 7 direct
xGMI links out of rank 0 carry their shards at the same time -- RCCL has no native
scatter, and xGMI is point-to-point, so this is the whole collective -- straight
into each rank's persistent input tensor, and the PCM (1 KiB per channel-block) is
gathered back the same way.

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


def _run(ops):
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def device_identity(index: int) -> dict:
    """What tells one GPU of a node from another, for the bench line's `rccl` block (the proof that N ranks ran on N
    devices): PCI bus id from the HIP runtime the process already holds, uuid and name from torch's properties."""
    import ctypes
    import socket
    out = {"host": socket.gethostname(), "local_device": int(index)}
    try:
        props = torch.cuda.get_device_properties(index)
        out["name"] = props.name
        if hasattr(props, "uuid"):
            out["uuid"] = str(props.uuid)
    except Exception as e:                                   # noqa: BLE001 (identity is best effort, the bus id below is the key)
        out["props_error"] = repr(e)
    try:
        from . import _lib
        rt = _lib._load_hip_runtime()
        buf = ctypes.create_string_buffer(64)
        if rt.hipDeviceGetPCIBusId(buf, 64, int(index)) == 0:
            out["pci_bus_id"] = buf.value.decode()
    except Exception as e:                                   # noqa: BLE001
        out["pci_error"] = repr(e)
    return out


def scatter_iq(iq_all: Optional[torch.Tensor], mine: torch.Tensor, n_channels: int, src: int = 0) -> torch.Tensor:
    """Rank `src` holds int8 [n_channels, blocks, block_bytes] (contiguous); every rank receives its shard INTO
    `mine` ([hi - lo, blocks, block_bytes], contiguous, persistent: no allocation, no second copy per step).
    Shards may differ by one channel, hence sends and receives rather than a padded collective."""
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = channel_range(rank, world, n_channels)
    assert mine.shape[0] == hi - lo and mine.is_contiguous()
    # gloo carries host tensors only: device shards are staged through the host (REHEARSALS of bench.py --gpus N on a
    # one-GPU box, tests; under nccl = RCCL the device tensors go as they are, over xGMI)
    staged = mine.is_cuda and dist.get_backend() == "gloo"
    if rank == src:
        assert iq_all is not None and iq_all.is_contiguous()
        ops = []
        for r in range(world):
            a, b = channel_range(r, world, n_channels)
            if r == src:
                mine.copy_(iq_all[a:b])
            elif b > a:
                ops.append(dist.P2POp(dist.isend, iq_all[a:b].cpu() if staged else iq_all[a:b], r))   # a slice of leading rows is contiguous
        _run(ops)
    elif hi > lo:
        if staged:
            host = torch.empty(mine.shape, dtype=mine.dtype)
            _run([dist.P2POp(dist.irecv, host, src)])
            mine.copy_(host)
        else:
            _run([dist.P2POp(dist.irecv, mine, src)])
    return mine


def gather_pcm(pcm_mine: torch.Tensor, out: Optional[torch.Tensor], n_channels: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Inverse of scatter_iq for the int16 [shard, blocks, n_pcm] output: rank dst receives every shard into
    `out` ([n_channels, blocks, n_pcm], persistent); the others pass out=None."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if world == 1:
        if out is not None:
            out.copy_(pcm_mine)
            return out
        return pcm_mine
    if rank == dst:
        assert out is not None and out.is_contiguous()
        ops = []
        for r in range(world):
            a, b = channel_range(r, world, n_channels)
            if r == dst:
                out[a:b].copy_(pcm_mine)
            elif b > a:
                ops.append(dist.P2POp(dist.irecv, out[a:b], r))
        _run(ops)
        return out
    if pcm_mine.shape[0] > 0:
        _run([dist.P2POp(dist.isend, pcm_mine.contiguous(), dst)])
    return None


def max_over_ranks(seconds: float, device: torch.device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
