"""The configurations of BASELINE.json at their full sizes, through the C ABI on the GPU:

  config 2 / 4 / north star   256, 512 (config 4's per-GPU shard) and 1024 (>= 1000, the stated target)
                              concurrent WBFM channels x 16 blocks in ONE launch
  config 3                    64 AM + 64 FM + 64 WBFM + 64 SSB x 16 blocks, one launch of k_rx_flow_bank
  config 5                    1024 SSB modulators

Each against the sequential CPU oracle on a spread of channels, plus size-independent properties over
ALL channels: channels fed identical input give identical output, every launch committed (nothing was
replayed), and the batch kernel (k_rx_wbfm_flow) and the block kernel (k_rx_wbfm) agree; a soak over
60 launches of random input with gates closing at random."""
import os
import zlib

import numpy as np
import pytest

from hackrfdiags_amd import api, synth
from tests.reflib import AM, FM, LSB, WBFM

pytestmark = pytest.mark.gpu
BLK = synth.BLOCK_BYTES
NBASE = 8


def _oracle_pcm(oracle, mode, x):
    o = oracle.rx()
    o.set_mode(mode)
    outs = [o.process(x[b]) for b in range(x.shape[0])]
    return np.stack([w[0] for w in outs]), [w[1] for w in outs]


@pytest.mark.parametrize("C", [512, 1024])
def test_wbfm_512_and_1024_channels(oracle, C):
    """C x 16 blocks of 262144 B per launch (2 and 4 GiB of IQ): the persistent grid is 2 and 4 workgroups
    per CU deep here.  Two launches (the second continues every stream)."""
    import torch
    B = 16
    dev = torch.device("cuda:0")
    base = [synth.make_input("fmtone" if k % 2 else "lcg", 500 + k, 2 * B).reshape(2 * B, BLK) for k in range(NBASE)]
    x = torch.empty((C, B, BLK), dtype=torch.int8, device=dev)
    want = [_oracle_pcm(oracle, WBFM, base[k]) for k in range(NBASE)]
    rx = api.Rx(C)
    rx.set_mode(api.WBFM)
    ref_other = api.Rx(C)
    ref_other.set_mode(api.WBFM)
    ref_other.debug_set_stream(0)                        # k_rx_wbfm: runs of blocks, phases in sequence
    for half in range(2):
        for c in range(C):
            x[c] = torch.from_numpy(base[c % NBASE][half * B:(half + 1) * B]).to(dev)
        out = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
        mag = torch.zeros((C, B), dtype=torch.int32, device=dev)
        npcm = torch.zeros((C, B), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        rx.process_device(x.data_ptr(), B * BLK, BLK, B, out.data_ptr(), d_n_pcm=npcm.data_ptr(), d_magnitude=mag.data_ptr())
        assert rx.sync() == 0
        got, gmag = out.cpu().numpy(), mag.cpu().numpy()
        assert int(npcm.sum().item()) == C * B * 512
        for k in range(NBASE):
            assert (got[k] == want[k][0][half * B:(half + 1) * B]).all(), (half, k)
            assert gmag[k].tolist() == want[k][1][half * B:(half + 1) * B], (half, k)
            for c in range(k, C, NBASE):                  # every channel with this input
                assert (got[c] == got[k]).all() and (gmag[c] == gmag[k]).all(), (half, c, k)
        out2 = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
        torch.cuda.synchronize()
        ref_other.process_device(x.data_ptr(), B * BLK, BLK, B, out2.data_ptr())
        assert ref_other.sync() == 0
        assert zlib.crc32(out2.cpu().numpy().tobytes()) == zlib.crc32(got.tobytes())
    assert rx.debug_counters()[5] == 0 and ref_other.debug_counters()[5] == 0, "a launch was replayed"


def test_mixed_bank_full_size(oracle):
    """config 3: 64 AM + 64 FM + 64 WBFM + 64 SSB channels x 16 blocks, two launches"""
    import torch
    C, B = 256, 16
    modes = [AM, FM, WBFM, LSB]
    amodes = [api.AM, api.FM, api.WBFM, api.LSB]
    dev = torch.device("cuda:0")
    base = [synth.make_input(("fmtone", "lcg")[k], 600 + k, 2 * B).reshape(2 * B, BLK) for k in range(2)]
    rx = api.Rx(C)
    for c in range(C):
        rx.set_mode(amodes[(4 * c) // C], channel=c)
    want = {(q, k): _oracle_pcm(oracle, modes[q], base[k]) for q in range(4) for k in range(2)}
    x = torch.empty((C, B, BLK), dtype=torch.int8, device=dev)
    for half in range(2):
        for c in range(C):
            x[c] = torch.from_numpy(base[c % 2][half * B:(half + 1) * B]).to(dev)
        out = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
        torch.cuda.synchronize()
        rx.process_device(x.data_ptr(), B * BLK, BLK, B, out.data_ptr())
        assert rx.sync() == 0
        got = out.cpu().numpy()
        for q in range(4):
            c0 = q * (C // 4)
            for k in range(2):
                assert (got[c0 + k] == want[(q, k)][0][half * B:(half + 1) * B]).all(), (half, q, k)
                for c in range(c0 + k, c0 + C // 4, 2):
                    assert (got[c] == got[c0 + k]).all(), (half, q, c)
    assert rx.debug_counters()[5] == 0


def test_ssb_modulator_1024_channels(oracle):
    """config 5: 1024 SSB modulators, 16 blocks of 512 PCM samples each, two calls"""
    C, B = 1024, 16
    base = [synth.lcg_pcm(7 + k, 2 * B * 512) for k in range(NBASE)]
    pcm = np.stack([base[c % NBASE] for c in range(C)])
    m = api.Mod(api.MOD_SSB, C)
    for c in range(C):
        m.set_sideband(c % 16 < 8, channel=c)            # channels k and k + 8 differ in sideband only
    os_ = {}
    for k in range(2 * NBASE):
        o = oracle.ssbmod(k < NBASE)
        os_[k] = o
    for half in range(2):
        seg = pcm[:, half * B * 512:(half + 1) * B * 512]
        got = m.process(seg)
        for k in range(2 * NBASE):
            want = np.concatenate([os_[k].process(base[k % NBASE][half * B * 512 + s:half * B * 512 + s + 512]) for s in range(0, B * 512, 512)])
            assert (got[k] == want).all(), (half, k)
            for c in range(k, C, 16):
                assert (got[c] == got[k]).all(), (half, c, k)



def test_soak_flow_shapes_agree_launch_after_launch():
    """Many full-size launches in a row, streams continuing from launch to launch: 256 channels of all four modes x 16
    blocks of fresh random input per launch, a real squelch threshold, and per-block levels drawn at random so that gates
    close and reopen all over the bank (one block in five is a whisper under the threshold).  Two handles demodulate the
    same input -- the one-launch bank kernel (k_rx_flow_bank) and one flow kernel per mode, each with its gated passes
    behind it -- and must agree on every PCM sample, n_pcm, magnitude and signal_allowed of every launch, with nothing
    left uncommitted.  (The oracle is not in this one: the PCM of both shapes is pinned to it by the other tests, gates
    included; this is about the flow kernels' flag protocol and the device-side repair holding up under load, launch
    after launch, on input nobody chose.)"""
    import torch
    C, B, launches = 256, 16, int(os.environ.get("HRFD_SOAK_LAUNCHES", "300"))   # (a long soak: HRFD_SOAK_LAUNCHES=2000)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(20260)
    handles = []
    for shape in (-1, 2):                                 # the bank kernel; one flow kernel per kind
        rx = api.Rx(C)
        for c in range(C):
            rx.set_mode([api.AM, api.FM, api.WBFM, api.LSB][c % 4], channel=c)
        rx.set_threshold(-22)
        rx.debug_set_fir_flow(shape)
        handles.append(rx)
    outs = [[torch.zeros((C, B, 512), dtype=torch.int16, device=dev), torch.zeros((C, B), dtype=torch.int32, device=dev),
             torch.zeros((C, B), dtype=torch.int32, device=dev), torch.zeros((C, B), dtype=torch.uint8, device=dev)] for _ in handles]
    shift_of = torch.tensor([0, 2, 4, 7, 0], device=dev)
    closed = 0
    for it in range(launches):
        x = torch.randint(-128, 128, (C, B, BLK), dtype=torch.int8, device=dev, generator=g)
        lvl = torch.randint(0, 5, (C, B, 1), device=dev, generator=g)
        x = (x.to(torch.int16) >> shift_of[lvl].to(torch.int16)).to(torch.int8)   # full scale, -12 dB, -24 dB, a whisper, full scale
        for o in outs:
            for t in o:
                t.fill_(77)
        torch.cuda.synchronize()
        for rx, o in zip(handles, outs):
            rx.process_device(x.data_ptr(), B * BLK, BLK, B, o[0].data_ptr(), d_n_pcm=o[1].data_ptr(), d_magnitude=o[2].data_ptr(),
                              d_allowed=o[3].data_ptr())
        assert handles[0].sync() == 0 and handles[1].sync() == 0, it
        closed += int((outs[0][3] == 0).sum().item())
        for a, b in zip(outs[0], outs[1]):
            assert torch.equal(a, b), it
    assert closed > launches * C                          # gates did close, all over the bank


def test_soak_wbfm_flow_against_the_block_kernel():
    """The same over the two WBFM kernels: 256 channels x 16 blocks of fresh random input per launch through
    k_rx_wbfm_flow (one continuous stream per channel, tiles verified and repaired in the service waves) and through the
    block kernel k_rx_wbfm (runs of blocks per workgroup, barriers between its phases) -- two implementations of the
    speculation that share little code.  PCM and magnitudes identical launch after launch, everything committed."""
    import torch
    C, B, launches = 256, 16, int(os.environ.get("HRFD_SOAK_LAUNCHES", "150"))
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(777)
    flow, block = api.Rx(C), api.Rx(C)
    for rx in (flow, block):
        rx.set_mode(api.WBFM)
    block.debug_set_stream(0)
    outs = [[torch.zeros((C, B, 512), dtype=torch.int16, device=dev), torch.zeros((C, B), dtype=torch.int32, device=dev)] for _ in range(2)]
    for it in range(launches):
        x = torch.randint(-128, 128, (C, B, BLK), dtype=torch.int8, device=dev, generator=g)
        if it % 3 == 1:
            x[:, :, ::2] = (x[:, :, ::2].to(torch.int16) >> 3).to(torch.int8)      # a weak I rail: other octants, other table rows
        torch.cuda.synchronize()
        for rx, o in zip((flow, block), outs):
            rx.process_device(x.data_ptr(), B * BLK, BLK, B, o[0].data_ptr(), d_magnitude=o[1].data_ptr())
        assert flow.sync() == 0 and block.sync() == 0, it
        for a, b in zip(outs[0], outs[1]):
            assert torch.equal(a, b), it


@pytest.mark.parametrize("kind", ["wbfm", "mixed"])
def test_realtime_cadence_1024_channels_one_block_per_batch(oracle, kind):
    """The north star's target in the reference's own cadence (bench.py `also.realtime_1024x1`): 1024 channels, ONE
    262144-byte block per channel per batch (hackRf/hackrf.c:100-101, DataConsumer.cc:219-262), from pinned host memory
    through hrfd_ingest_*, PCM back on the host, batch after batch with the streams continuing.  Four batches; PCM,
    magnitude and gate of a spread of channels against the sequential oracle, and every channel against the channel
    that was fed the same input."""
    C, NB = 1024, 4
    modes = [AM, FM, WBFM, LSB]
    amodes = [api.AM, api.FM, api.WBFM, api.LSB]
    base = [synth.make_input("fmtone" if k % 2 else "lcg", 900 + k, NB).reshape(NB, BLK) for k in range(NBASE)]
    rx = api.Rx(C)
    if kind == "mixed":
        for c in range(C):
            rx.set_mode(amodes[(4 * c) // C], channel=c)
    else:
        rx.set_mode(api.WBFM)
    ing = api.Ingest(rx, BLK, 1, 2)
    got = []
    for t in range(NB):
        slot = ing.acquire()
        for k in range(NBASE):
            slot[k::NBASE, 0] = base[k][t]
        ing.submit(0)
        got.append(ing.collect())
    assert ing.replayed() == 0
    ing.close()
    quarters = range(4) if kind == "mixed" else [2]
    for q in quarters:
        lo, hi = (q * C // 4, (q + 1) * C // 4) if kind == "mixed" else (0, C)
        for k in range(NBASE):
            o = oracle.rx()
            o.set_mode(modes[q])
            first = lo + ((k - lo) % NBASE)              # the first channel of this quarter with base input k
            for t in range(NB):
                p, m, a, _ = o.process(base[k][t])
                pcm, n_pcm, mag, allowed = got[t]
                assert n_pcm[first, 0] == 512 and bool(allowed[first, 0]) == a and int(mag[first, 0]) == m, (q, k, t)
                assert (pcm[first, 0] == p).all(), (q, k, t)
                same = np.arange(first, hi, NBASE)
                assert (pcm[same, 0] == pcm[first, 0]).all() and (mag[same, 0] == mag[first, 0]).all(), (q, k, t)
