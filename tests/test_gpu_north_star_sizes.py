"""The configurations of BASELINE.json at their full sizes, through the C ABI on the GPU:

  config 2 / 4 / north star   256, 512 (config 4's per-GPU shard) and 1024 (>= 1000, the stated target)
                              concurrent WBFM channels x 16 blocks in ONE launch
  config 3                    64 AM + 64 FM + 64 WBFM + 64 SSB x 16 blocks, one launch of k_rx_flow_bank
  config 5                    1024 SSB modulators

Every channel is fed an input of ITS OWN (round 5; tests/fullsize.py says why: until round 4 the inputs had period 8,
the number of XCDs, and a channel mapped onto c +- 8k would have gone unseen): 64 channels drawn from the whole range
-- 0, C - 1, every residue mod 8 -- against the sequential CPU oracle, plus size-independent properties over ALL
channels: the batch kernel (k_rx_wbfm_flow) and the block kernel (k_rx_wbfm) agree on every channel's distinct input,
channels fed identical input (period 7, coprime to 8) give identical output, every launch committed (nothing was
replayed); soaks over hundreds of launches of random input with gates closing at random."""
import os

import numpy as np
import pytest

from hackrfdiags_amd import api, synth
from tests.fullsize import PERIOD, check_rx_bank_distinct, check_rx_bank_period, distinct_batch, oracle_rx_stream, pick_channels
from tests.reflib import AM, FM, LSB, WBFM

pytestmark = pytest.mark.gpu
BLK = synth.BLOCK_BYTES


@pytest.mark.parametrize("C", [512, 1024])
def test_wbfm_512_and_1024_channels(oracle, C):
    """C x 16 blocks of 262144 B per launch (2 and 4 GiB of IQ): the persistent grid is 2 and 4 workgroups
    per CU deep here.  Two launches (the second continues every stream).  EVERY channel has an input of its own
    (tests/fullsize.py); 64 channels drawn from the whole range (0, C - 1, every residue mod 8) against the sequential
    oracle, and the block kernel k_rx_wbfm (runs of blocks, phases in sequence) must agree on ALL channels."""
    check_rx_bank_distinct(oracle, api, C, 16, lambda c: WBFM, twin=lambda rx2: rx2.debug_set_stream(0))


def test_wbfm_256_channels_batches_of_48_blocks(oracle):
    """batches longer than 16 blocks: since round 5 a workgroup streams up to 64 blocks of its channel as ONE run (one table
    copy, no re-derived history, one service tail; rounds 2-4: runs of 16).  256 channels x 48 blocks (3 GiB) per launch,
    two launches, every channel an input of its own; 48 channels of the whole range against the oracle, the block kernel
    agreeing on all 256."""
    check_rx_bank_distinct(oracle, api, 256, 48, lambda c: WBFM, n_check=48, twin=lambda rx2: rx2.debug_set_stream(0))


def test_wbfm_1024_channels_all_channels_period_7(oracle):
    """the all-channel property at 1024 channels with a period coprime to the 8 XCDs: equal input => equal output"""
    check_rx_bank_period(oracle, api, 1024, 16, lambda c: WBFM, seed=500)


MIXED = [AM, FM, WBFM, LSB]


def test_mixed_bank_full_size(oracle):
    """config 3: 64 AM + 64 FM + 64 WBFM + 64 SSB channels x 16 blocks, two launches, every channel an input of its
    own; 64 channels of the whole bank (all four quarters) against the oracle in their modes; then the all-channel
    property with period 7"""
    C = 256
    check_rx_bank_distinct(oracle, api, C, 16, lambda c: MIXED[(4 * c) // C])
    check_rx_bank_period(oracle, api, C, 16, lambda c: MIXED[(4 * c) // C], seed=600)


def test_ssb_modulator_1024_channels(oracle):
    """config 5: 1024 SSB modulators, 16 blocks of 512 PCM samples each, two calls.  Every channel has PCM of its own
    (LCG seed 7 + c) and a sideband that does not follow the channel number mod 8; 64 channels of the whole
    range against the oracle's modulator.  Then the all-channel property with period 7: equal (input, sideband) =>
    equal bytes."""
    C, B = 1024, 16
    n = 2 * B * 512
    lsb_of = lambda c: (c % 3) != 0                       # noqa: E731
    # --- distinct inputs
    pcm = np.stack([synth.lcg_pcm(7 + c, n) for c in range(C)])
    assert len({pcm[c, :64].tobytes() for c in range(C)}) == C
    m = api.Mod(api.MOD_SSB, C)
    for c in range(C):
        m.set_sideband(lsb_of(c), channel=c)
    sel = pick_channels(C)
    orc = {c: oracle.ssbmod(lsb_of(c)) for c in sel}
    for half in range(2):
        got = m.process(pcm[:, half * B * 512:(half + 1) * B * 512])
        for c in sel:
            want = np.concatenate([orc[c].process(pcm[c, half * B * 512 + s:half * B * 512 + s + 512]) for s in range(0, B * 512, 512)])
            assert (got[c] == want).all(), (half, c)
        del got
    del m
    # --- period 7, every channel
    base = [synth.lcg_pcm(7007 + k, n) for k in range(PERIOD)]
    pcm = np.stack([base[c % PERIOD] for c in range(C)])
    side = lambda c: (c // PERIOD) % 2 == 0               # noqa: E731
    m = api.Mod(api.MOD_SSB, C)
    for c in range(C):
        m.set_sideband(side(c), channel=c)
    firsts = {}
    for c in range(C):
        firsts.setdefault((c % PERIOD, side(c)), c)
    orc = {key: oracle.ssbmod(key[1]) for key in firsts}
    for half in range(2):
        got = m.process(pcm[:, half * B * 512:(half + 1) * B * 512])
        for key, c0 in firsts.items():
            want = np.concatenate([orc[key].process(base[key[0]][half * B * 512 + s:half * B * 512 + s + 512]) for s in range(0, B * 512, 512)])
            assert (got[c0] == want).all(), (half, key)
        for c in range(C):
            c0 = firsts[(c % PERIOD, side(c))]
            assert c == c0 or (got[c] == got[c0]).all(), (half, c, c0)
        del got


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
    through hrfd_ingest_*, PCM back on the host, batch after batch with the streams continuing.  Four batches, every
    channel an input of its own: PCM, magnitude and gate of 64 channels of the whole range against the sequential
    oracle.  Then four more batches on a fresh bank with inputs of period 7: every channel against the channel of its
    mode that was fed the same input."""
    import torch
    C, NB = 1024, 4
    mode_of = (lambda c: MIXED[(4 * c) // C]) if kind == "mixed" else (lambda c: WBFM)

    def bank():
        rx = api.Rx(C)
        if kind == "mixed":
            for c in range(C):
                rx.set_mode(mode_of(c), channel=c)
        else:
            rx.set_mode(api.WBFM)
        return rx, api.Ingest(rx, BLK, 1, 2)

    def run(ing, x):
        got = []
        for t in range(NB):
            slot = ing.acquire()
            slot[:, 0] = x[:, t]
            ing.submit(0)
            got.append(ing.collect())
        assert ing.replayed() == 0
        ing.close()
        return got

    # --- distinct inputs
    x = distinct_batch(C, NB, torch.device("cuda:0")).cpu().numpy()
    torch.cuda.empty_cache()
    rx, ing = bank()
    got = run(ing, x)
    for c in pick_channels(C):
        for t, (p, m, a) in enumerate(oracle_rx_stream(oracle, mode_of(c), x[c])):
            pcm, n_pcm, mag, allowed = got[t]
            assert n_pcm[c, 0] == 512 and bool(allowed[c, 0]) == a and int(mag[c, 0]) == m, (c, t)
            assert (pcm[c, 0] == p).all(), (c, t)
    del x
    # --- period 7, every channel
    base = np.stack([synth.make_input("fmtone" if k % 2 else "lcg", 900 + k, NB).reshape(NB, BLK) for k in range(PERIOD)])
    rx, ing = bank()
    got = run(ing, base[np.arange(C) % PERIOD])
    firsts = {}
    for c in range(C):
        firsts.setdefault((mode_of(c), c % PERIOD), c)
    for (mode, r), c0 in firsts.items():
        for t, (p, m, a) in enumerate(oracle_rx_stream(oracle, mode, base[r])):
            assert (got[t][0][c0, 0] == p).all() and int(got[t][2][c0, 0]) == m, (mode, r, t)
    for t in range(NB):
        pcm, n_pcm, mag, allowed = got[t]
        for c in range(C):
            c0 = firsts[(mode_of(c), c % PERIOD)]
            assert c == c0 or ((pcm[c, 0] == pcm[c0, 0]).all() and mag[c, 0] == mag[c0, 0]), (t, c, c0)
