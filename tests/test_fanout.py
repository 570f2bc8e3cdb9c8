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


def _fanout_bank(oracle, C, B, devices, launches, n_check, period_too):
    """C WBFM channels x B blocks per batch through hrfd_fanout_* over `devices` (shard g on devices[g]), the IQ of the
    whole bank on device 0, EVERY channel an input of its own: `n_check` channels of the whole range against the
    sequential oracle, ALL channels against one handle over the whole bank on device 0 (the same device code, another
    partition: a shard that reads or writes another shard's rows shows), nothing replayed.  `period_too`: one more
    batch on a fresh fan-out with inputs of period 7 -- every channel against the channel fed the same input."""
    import torch
    from tests.fullsize import PERIOD, distinct_batch, assert_all_distinct, oracle_rx_stream, pick_channels
    dev = torch.device("cuda:0")
    nsh = len(devices)
    assert sum(api.fanout_channel_range(C, nsh, g)[1] for g in range(nsh)) == C
    x = distinct_batch(C, launches * B, dev)
    assert_all_distinct(x)
    sel = pick_channels(C, n_check)
    tsel = torch.tensor(sel, device=dev)
    fo = api.Fanout(C, devices)
    fo.set_mode(api.WBFM)
    one = api.Rx(C)
    one.set_mode(api.WBFM)
    out = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
    npcm = torch.zeros((C, B), dtype=torch.int32, device=dev)
    got = []
    s = torch.cuda.Stream()
    for k in range(launches):
        with torch.cuda.stream(s):
            xs = x[:, k * B:(k + 1) * B].contiguous()
        fo.scatter(0, xs.data_ptr(), BLK, B, src_stream=s.cuda_stream)
        fo.process(0)
        assert fo.collect(0, out.data_ptr(), npcm.data_ptr()) == 0, "a channel was replayed"
        torch.cuda.synchronize()
        assert int(npcm.sum().item()) == C * B * 512
        got.append(out[tsel].cpu().numpy())
        out1 = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
        torch.cuda.synchronize()
        one.process_device(xs.data_ptr(), B * BLK, BLK, B, out1.data_ptr())
        assert one.sync() == 0
        assert torch.equal(out1, out), k                  # every channel, on the device
        del xs, out1
    xsel = x[tsel].cpu().numpy()
    del x
    one.close()
    torch.cuda.empty_cache()
    for i, c in enumerate(sel):
        for b, (p, _, _) in enumerate(oracle_rx_stream(oracle, WBFM, xsel[i])):
            assert (got[b // B][i, b % B] == p).all(), (c, b)
    fo.close()
    if period_too:
        base = np.stack([synth.make_input("fmtone" if k % 2 else "lcg", 800 + k, B).reshape(B, BLK) for k in range(PERIOD)])
        fo = api.Fanout(C, devices)
        fo.set_mode(api.WBFM)
        idx = torch.arange(C, device=dev) % PERIOD
        with torch.cuda.stream(s):
            xs = torch.from_numpy(base).to(dev)[idx]
        fo.scatter(0, xs.data_ptr(), BLK, B, src_stream=s.cuda_stream)
        fo.process(0)
        assert fo.collect(0, out.data_ptr(), npcm.data_ptr()) == 0
        torch.cuda.synchronize()
        del xs
        assert bool((out[PERIOD:] == out[idx[PERIOD:]]).all())          # channel c = channel c mod 7, on the device
        head = out[:PERIOD].cpu().numpy()
        for k in range(PERIOD):
            for b, (p, _, _) in enumerate(oracle_rx_stream(oracle, WBFM, base[k])):
                assert (head[k, b] == p).all(), (k, b)
        fo.close()
    torch.cuda.empty_cache()


@pytest.mark.gpu
def test_config4_full_size_through_the_fanout_on_one_device(oracle):
    """BASELINE config 4 as far as one GPU allows: 4096 WBFM channels x 16 blocks (16 GiB of IQ on the source device)
    through hrfd_fanout_* with EIGHT shards of 512 channels -- all eight on device 0 here; on an 8-GPU node the same
    calls put one shard on each GPU and the scatter's copies go over xGMI (test_fanout_over_the_devices_of_a_node,
    below, switches itself on there).  Two batches (every stream continues), every channel an input of its own."""
    assert [api.fanout_channel_range(4096, 8, g) for g in range(8)] == [(512 * g, 512) for g in range(8)]
    _fanout_bank(oracle, 4096, 16, [0] * 8, launches=2, n_check=64, period_too=True)


@pytest.mark.gpu
@pytest.mark.parametrize("n_dev", [2, 3, 4, 8])
def test_fanout_over_the_devices_of_a_node(oracle, n_dev):
    """The same bank with one shard per DEVICE: hipDeviceEnablePeerAccess, per-device table replication and the
    hipMemcpyPeerAsync scatter / gather of hrfd_fanout.hip over xGMI.  Needs n_dev devices: skipped on the one-GPU
    boxes this repository has been developed on (never run so far), switched on by itself on a node."""
    if api.device_count() < n_dev:
        pytest.skip(f"needs {n_dev} devices, this box has {api.device_count()}")
    _fanout_bank(oracle, 512 * n_dev, 16, list(range(n_dev)), launches=2, n_check=64, period_too=True)
