"""world_size-2 (and 3) gloo tests of the multi-GPU plumbing: channel sharding,
the rank-0 scatter of IQ blocks, the PCM gather and the max-over-ranks timing.
The compute in the middle is a stand-in checksum: the HIP path itself needs a GPU
and is covered by the -m gpu tests."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hackrfdiags_amd import shard


def test_channel_range_partitions_exactly():
    for world in (1, 2, 3, 8):
        for n in (1, 7, 8, 256, 4096, 1000):
            got = [shard.channel_range(r, world, n) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(got[i][1] == got[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in got]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, n_channels, blocks, block_bytes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    torch.manual_seed(0)
    iq_all = None
    ref = torch.randint(-128, 128, (n_channels, blocks, block_bytes), dtype=torch.int8)   # same on all ranks
    if rank == 0:
        iq_all = ref.clone()
    mine = shard.scatter_iq(iq_all, n_channels, blocks, block_bytes, dev)
    lo, hi = shard.channel_range(rank, world, n_channels)
    ok = bool((mine == ref[lo:hi]).all())
    # stand-in for the demodulator: 4 "PCM" values per channel-block derived from the shard
    pcm = mine.to(torch.int16).reshape(hi - lo, blocks, 4, -1).sum(dim=3).to(torch.int16)
    allpcm = shard.gather_pcm(pcm, n_channels)
    if rank == 0:
        want = ref.to(torch.int16).reshape(n_channels, blocks, 4, -1).sum(dim=3).to(torch.int16)
        ok = ok and bool((allpcm == want).all())
    t = shard.max_over_ranks(0.25 * (rank + 1), dev)
    ok = ok and abs(t - 0.25 * world) < 1e-9
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


@pytest.mark.parametrize("world,n_channels", [(2, 8), (2, 5), (3, 7)])
def test_scatter_process_gather_gloo(world, n_channels):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + world * 7 + n_channels
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_channels, 2, 4096, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    results = dict(q.get(timeout=10) for _ in range(world))
    assert all(results[r] for r in range(world)), results
