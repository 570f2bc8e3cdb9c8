"""world_size-2 (and 3) gloo tests of the multi-GPU path: channel sharding, the grouped scatter of IQ blocks out
of rank 0 into every rank's persistent input tensor, the demodulation of each rank's shard, the PCM gather and
the max-over-ranks timing.  The demodulator in the middle is the real chain -- the CPU oracle, since this container
has no GPU (on a GPU box the ranks run libhrfd on their shards exactly where the oracle runs here; the -m gpu
tests pin libhrfd to the oracle) -- and rank 0 checks the gathered PCM against a single-process oracle run."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hackrfdiags_amd import shard, synth


def test_channel_range_partitions_exactly():
    for world in (1, 2, 3, 8):
        for n in (1, 7, 8, 256, 4096, 1000):
            got = [shard.channel_range(r, world, n) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(got[i][1] == got[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in got]
            assert max(sizes) - min(sizes) <= 1


def _oracle_pcm(xs):
    """[channels, blocks, bytes] int8 -> [channels, blocks, 512] int16, WBFM, one sequential oracle per channel.
    (The ranks only LOAD the oracle: the parent test builds it once before spawning, so that concurrent ranks
    never race on writing the same .so.)"""
    from tests.reflib import Oracle, WBFM
    orc = Oracle()
    out = np.zeros((xs.shape[0], xs.shape[1], 512), dtype=np.int16)
    for c in range(xs.shape[0]):
        o = orc.rx()
        o.set_mode(WBFM)
        for b in range(xs.shape[1]):
            out[c, b] = o.process(xs[c, b])[0]
    return out


def _inputs(n_channels, blocks):
    return np.stack([synth.make_input("fmtone" if c % 2 else "lcg", 700 + c, blocks).reshape(blocks, synth.BLOCK_BYTES)
                     for c in range(n_channels)])


def _worker(rank, world, port, n_channels, blocks, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    BLK = synth.BLOCK_BYTES
    ref = _inputs(n_channels, blocks)                       # every rank can build it: rank 0 is the one that sends
    lo, hi = shard.channel_range(rank, world, n_channels)
    mine = torch.zeros((hi - lo, blocks, BLK), dtype=torch.int8)     # persistent input tensor of this rank
    iq_all = torch.from_numpy(ref).contiguous() if rank == 0 else None
    allpcm = torch.zeros((n_channels, blocks, 512), dtype=torch.int16) if rank == 0 else None
    ok = True
    for step in range(2):                                   # twice through the same buffers: nothing is reallocated
        ptr = mine.data_ptr()
        shard.scatter_iq(iq_all, mine, n_channels)
        ok = ok and mine.data_ptr() == ptr and bool((mine.numpy() == ref[lo:hi]).all())
        pcm = torch.from_numpy(_oracle_pcm(mine.numpy()))    # the rank's demodulator on its shard
        shard.gather_pcm(pcm, allpcm, n_channels)
    if rank == 0:
        ok = ok and bool((allpcm.numpy() == _oracle_pcm(ref)).all())
    t = shard.max_over_ranks(0.25 * (rank + 1), dev)
    ok = ok and abs(t - 0.25 * world) < 1e-9
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


@pytest.mark.parametrize("world,n_channels", [(2, 4), (2, 5), (3, 7)])
def test_scatter_demodulate_gather_gloo(world, n_channels):
    from tests.reflib import build_oracle
    build_oracle()                                          # once, here: the ranks find it up to date
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + world * 7 + n_channels
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_channels, 2, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    results = dict(q.get(timeout=10) for _ in range(world))
    assert all(results[r] for r in range(world)), results


# ---------------------------------------------------------------------------------------------------------------
# The same path with libhrfd in the middle (GPU box).  A one-GPU box cannot run RCCL between two ranks (RCCL refuses
# two ranks on one device), so the ranks meet over gloo with host-staged shards -- rendezvous, sharding, scatter,
# every rank's hrfd_rx on its own channels with its own resident state, gather, MAX over ranks are the code of
# bench.py --gpus N / shard.py -- and share GPU 0.  What a multi-GPU node adds is the transport (RCCL instead of
# gloo) and one device per rank: unmeasured over xGMI until the driver has such a node.
# ---------------------------------------------------------------------------------------------------------------
def _gpu_worker(rank, world, port, n_channels, blocks, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hackrfdiags_amd import api
    BLK = synth.BLOCK_BYTES
    lo, hi = shard.channel_range(rank, world, n_channels)
    ref = _inputs(n_channels, 2 * blocks) if rank == 0 else None
    mine = torch.zeros((hi - lo, blocks, BLK), dtype=torch.int8)
    allpcm = torch.zeros((n_channels, blocks, 512), dtype=torch.int16) if rank == 0 else None
    rx = api.Rx(hi - lo, device=0)                          # this rank's shard: its channels' state lives here for the whole stream
    rx.set_mode(api.WBFM)
    ok = True
    got = []
    for step in range(2):                                   # two consecutive batches: the shards' streams continue
        iq_all = torch.from_numpy(ref[:, step * blocks:(step + 1) * blocks].copy()).contiguous() if rank == 0 else None
        shard.scatter_iq(iq_all, mine, n_channels)
        pcm = rx.process_block(mine.numpy(), blocks)[0]      # libhrfd on the rank's shard (flow kernel: blocks > 1)
        shard.gather_pcm(torch.from_numpy(np.ascontiguousarray(pcm)), allpcm, n_channels)
        if rank == 0:
            got.append(allpcm.numpy().copy())
    if rank == 0:
        one = api.Rx(n_channels, device=0)                  # ONE handle over the whole bank, same batches
        one.set_mode(api.WBFM)
        for step in range(2):
            want = one.process_block(ref[:, step * blocks:(step + 1) * blocks], blocks)[0]
            ok = ok and bool((got[step] == want).all())
        full = _oracle_pcm(ref)                              # and the sequential CPU oracle
        ok = ok and bool((np.concatenate(got, axis=1) == full).all())
    t = shard.max_over_ranks(0.25 * (rank + 1), torch.device("cpu"))
    ok = ok and abs(t - 0.25 * world) < 1e-9
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


@pytest.mark.gpu
@pytest.mark.parametrize("world,n_channels", [(2, 6), (3, 7)])
def test_ranks_run_libhrfd_on_their_shards(world, n_channels):
    from tests.reflib import build_oracle
    build_oracle()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000 + world * 7 + n_channels
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, n_channels, 3, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    results = dict(q.get(timeout=10) for _ in range(world))
    assert all(results[r] for r in range(world)), results


# ---------------------------------------------------------------------------------------------------------------
# The nccl (= RCCL) twin: one rank per DEVICE, device tensors end to end, the grouped scatter / gather over xGMI --
# the path bench.py --gpus N takes on a node.  Needs `world` devices: skipped on the one-GPU boxes this repository
# has been developed on (never run so far); it switches itself on the day a node is there.
# ---------------------------------------------------------------------------------------------------------------
def _nccl_worker(rank, world, port, n_channels, blocks, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from hackrfdiags_amd import api
    BLK = synth.BLOCK_BYTES
    lo, hi = shard.channel_range(rank, world, n_channels)
    ref = _inputs(n_channels, 2 * blocks) if rank == 0 else None
    mine = torch.zeros((hi - lo, blocks, BLK), dtype=torch.int8, device=dev)
    pcm = torch.zeros((hi - lo, blocks, 512), dtype=torch.int16, device=dev)
    allpcm = torch.zeros((n_channels, blocks, 512), dtype=torch.int16, device=dev) if rank == 0 else None
    rx = api.Rx(hi - lo, device=rank)
    rx.set_mode(api.WBFM)
    # proof that `world` ranks sit on `world` different devices
    ident = [None] * world
    dist.all_gather_object(ident, str(torch.cuda.get_device_properties(dev).uuid) if hasattr(torch.cuda.get_device_properties(dev), "uuid")
                           else f"{rank}:{torch.cuda.get_device_properties(dev).name}")
    ok = len(set(ident)) == world
    got = []
    for step in range(2):
        iq_all = torch.from_numpy(ref[:, step * blocks:(step + 1) * blocks].copy()).to(dev).contiguous() if rank == 0 else None
        shard.scatter_iq(iq_all, mine, n_channels)
        torch.cuda.synchronize()
        rx.process_device(mine.data_ptr(), blocks * BLK, BLK, blocks, pcm.data_ptr())
        ok = ok and rx.sync() == 0
        shard.gather_pcm(pcm, allpcm, n_channels)
        torch.cuda.synchronize()
        if rank == 0:
            got.append(allpcm.cpu().numpy().copy())
    if rank == 0:
        ok = ok and bool((np.concatenate(got, axis=1) == _oracle_pcm(ref)).all())
    t = shard.max_over_ranks(0.25 * (rank + 1), dev)
    ok = ok and abs(t - 0.25 * world) < 1e-9
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


@pytest.mark.gpu
@pytest.mark.parametrize("world,n_channels", [(2, 6), (4, 9), (8, 16)])
def test_ranks_run_libhrfd_on_their_own_devices_over_rccl(world, n_channels):
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} devices, this box has {torch.cuda.device_count()}")
    from tests.reflib import build_oracle
    build_oracle()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000 + world * 7 + n_channels
    procs = [ctx.Process(target=_nccl_worker, args=(r, world, port, n_channels, 3, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    results = dict(q.get(timeout=10) for _ in range(world))
    assert all(results[r] for r in range(world)), results
