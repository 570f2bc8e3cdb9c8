"""hrfd_fanout_*: one host process, several devices (the single-process wiring of Radio.cc:164-237 kept
for a host that drives a bank of channels on the GPUs of a node).  CPU: the shard arithmetic through the C ABI.
GPU (one device is enough: several shards may sit on one GPU): scatter from one source buffer, every shard
demodulating its channels, gather -- against ONE handle over the whole bank and against the oracle."""
import numpy as np
import pytest

from hackrfdiags_amd import api, shard, synth
from tests.reflib import WBFM, AM, FM, LSB

BLK = synth.BLOCK_BYTES


def test_fanout_channel_range_partitions_exactly():
    for world in (1, 2, 3, 7, 8):
        for n in (8, 9, 256, 4096, 1000, 1023):
            got = [api.fanout_channel_range(n, world, g) for g in range(world)]
            assert got[0][0] == 0 and got[-1][0] + got[-1][1] == n
            assert all(got[i][0] + got[i][1] == got[i + 1][0] for i in range(world - 1))
            sizes = [c for _, c in got]
            assert max(sizes) - min(sizes) <= 1
            # the same shards as the multi-process layer (hackrfdiags_amd/shard.py)
            assert [(lo, lo + cnt) for lo, cnt in got] == [shard.channel_range(g, world, n) for g in range(world)]
    with pytest.raises(api.HrfdError):
        api.fanout_channel_range(8, 2, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("n_shards,C", [(3, 7), (2, 8), (4, 9)])
def test_fanout_on_one_device_equals_one_handle_and_the_oracle(oracle, n_shards, C):
    import torch
    B = 3
    dev = torch.device("cuda:0")
    modes = [[WBFM, AM, FM, LSB][c % 4] if C == 9 else WBFM for c in range(C)]
    xs = np.stack([synth.make_input("fmtone" if c % 2 else "lcg", 300 + c, 2 * B).reshape(2 * B, BLK) for c in range(C)])
    xs[1, 1:3] = 0                                          # a gate that closes inside the first batch (threshold below)
    fo = api.Fanout(C, [0] * n_shards)
    one = api.Rx(C)
    for c in range(C):
        fo.set_mode(modes[c], channel=c)
        one.set_mode(modes[c], channel=c)
    fo.set_threshold(-30)
    one.set_threshold(-30)
    out = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
    npcm = torch.zeros((C, B), dtype=torch.int32, device=dev)
    s = torch.cuda.Stream()
    for half in range(2):
        with torch.cuda.stream(s):
            x = torch.from_numpy(xs[:, half * B:(half + 1) * B].copy()).to(dev, non_blocking=False)
        fo.scatter(0, x.data_ptr(), BLK, B, src_stream=s.cuda_stream)
        fo.process(0)
        fo.collect(0, out.data_ptr(), npcm.data_ptr())
        got, gn = out.cpu().numpy(), npcm.cpu().numpy()
        want = one.process_block(xs[:, half * B:(half + 1) * B], B)
        assert (gn == want[1]).all()
        assert (got == want[0]).all()
        for c in range(C):
            o = oracle.rx()
            o.set_mode(modes[c])
            o.set_threshold(-30)
            for b in range((half + 1) * B):
                p = o.process(xs[c, b])[0]
                if b >= half * B:
                    assert gn[c, b - half * B] == len(p) and (got[c, b - half * B, :len(p)] == p).all(), (half, c, b)


@pytest.mark.gpu
def test_config4_full_size_through_the_fanout_on_one_device(oracle):
    """BASELINE config 4 as far as one GPU allows: 4096 WBFM channels x 16 blocks (16 GiB of IQ on the source device)
    through hrfd_fanout_* with EIGHT shards of 512 channels -- all eight on device 0 here; on an 8-GPU node the same
    calls put one shard on each GPU and the scatter's copies go over xGMI (unmeasured until such a node exists).
    Two batches (every stream continues).  A spread of channels against the sequential oracle; every one of the 4096
    against the channel that was fed the same input; nothing replayed."""
    import torch
    C, B, NSH = 4096, 16, 8
    NBASE = 8
    dev = torch.device("cuda:0")
    base = np.stack([synth.make_input("fmtone" if k % 2 else "lcg", 800 + k, 2 * B).reshape(2 * B, BLK) for k in range(NBASE)])
    assert [api.fanout_channel_range(C, NSH, g) for g in range(NSH)] == [(512 * g, 512) for g in range(NSH)]
    fo = api.Fanout(C, [0] * NSH)
    fo.set_mode(api.WBFM)
    idx = (torch.arange(C, device=dev) % NBASE)
    out = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
    npcm = torch.zeros((C, B), dtype=torch.int32, device=dev)
    s = torch.cuda.Stream()
    for half in range(2):
        with torch.cuda.stream(s):
            bdev = torch.from_numpy(base[:, half * B:(half + 1) * B].copy()).to(dev)      # [NBASE, B, BLK]
            x = bdev[idx]                                                                  # [4096, 16, 262144]: 16 GiB
        fo.scatter(0, x.data_ptr(), BLK, B, src_stream=s.cuda_stream)
        fo.process(0)
        assert fo.collect(0, out.data_ptr(), npcm.data_ptr()) == 0, "a channel was replayed"
        torch.cuda.synchronize()
        del x
        assert int(npcm.sum().item()) == C * B * 512
        # every channel equals the first channel with its input, on the device (64 MiB of PCM)
        assert bool((out.view(C // NBASE, NBASE, B, 512) == out[:NBASE].unsqueeze(0)).all())
        got = out[:NBASE].cpu().numpy()
        for k in range(NBASE):
            o = oracle.rx()
            o.set_mode(WBFM)
            for b in range((half + 1) * B):
                p = o.process(base[k, b])[0]
                if b >= half * B:
                    assert (got[k, b - half * B] == p).all(), (half, k, b)
    fo.close()
    torch.cuda.empty_cache()
