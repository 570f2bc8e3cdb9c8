"""Parity tests proper: the HIP path (through the C ABI) against the committed
golden vectors and against the CPU oracle on seeded inputs.  Run on the GPU box
with `pytest -m gpu`."""
import numpy as np
import pytest

from hackrfdiags_amd import api, synth
from tests import goldencheck as G
from tests.reflib import AM, FM, WBFM, LSB, USB, NONE

pytestmark = pytest.mark.gpu
BLK = synth.BLOCK_BYTES
ARR, MAN = G.load()


@pytest.fixture(scope="module")
def engine():
    if api.device_count() < 1:
        pytest.fail("no GPU visible: the HIP path cannot run (and there is no CPU fallback)")
    return api.Engine()


# ---------------------------------------------------------------- golden vectors
@pytest.mark.parametrize("case", MAN["rx"], ids=lambda c: c["key"])
def test_golden_rx(engine, case):
    # bit-exact in every mode, the float recurrences (WBFM de-emphasis, AM/SSB dc
    # removal) included: tolerance 0 LSB where BASELINE.json would allow +-1
    G.check_rx_case(engine, ARR, case)


@pytest.mark.parametrize("case", MAN["frontend"], ids=lambda c: c["key"])
def test_golden_frontend(engine, case):
    G.check_frontend_case(engine, ARR, case)


@pytest.mark.parametrize("case", MAN["rx_long"], ids=lambda c: f"long_mode{c['mode']}")
def test_golden_rx_long(engine, case):
    G.check_long_case(engine, case)


def test_golden_squelch(engine):
    G.check_squelch(engine, ARR, MAN["squelch"][0])


@pytest.mark.parametrize("case", MAN["chunked"], ids=lambda c: c["key"])
def test_golden_chunked_1024(engine, oracle, case):
    """the reference is chunk-invariant for multiples of 64 bytes; the HIP path takes
    multiples of 1024: 256 calls of 1024 bytes must equal the 64-byte-chunk golden."""
    x = synth.make_input(case["kind"], case["seed"], 1)
    h = engine.rx()
    h.set_mode(case["mode"])
    got = np.concatenate([h.process(x[o:o + 1024])[0] for o in range(0, len(x), 1024)])
    assert (got == ARR[case["key"]]).all()


# ---------------------------------------------------------------- oracle parity, batched
def _oracle_stream(oracle, mode, x, nb, gain=None, threshold=None):
    o = oracle.rx()
    o.set_mode(mode)
    if gain is not None:
        o.set_gain(mode, gain)
    if threshold is not None:
        o.set_threshold(threshold)
    return [o.process(x[b]) for b in range(nb)]


@pytest.mark.parametrize("fir_flow", [0, 1], ids=["block_kernels", "flow_kernel_fir_modes"])
@pytest.mark.parametrize("mode", [AM, FM, LSB, USB])
@pytest.mark.parametrize("kind", ["lcg", "amtone", "dc_neg", "impulse"])
def test_batched_blocks_match_oracle_other_modes(oracle, mode, kind, fir_flow):
    """AM / FM / SSB batches on both implementations: one workgroup per channel-block (k_rx_fir, k_rx_post), and the
    flow kernel's FIR modes (one persistent workgroup per channel, the call as one stream: what a bank of 48 channels
    or more gets by default)"""
    C, B = 3, 4
    xs = np.stack([synth.make_input(kind, 30 + c, 2 * B) for c in range(C)]).reshape(C, 2 * B, BLK)
    rx = api.Rx(C)
    rx.set_mode(mode)
    rx.debug_set_fir_flow(fir_flow)
    r1 = rx.process_block(xs[:, :B], B)
    r2 = rx.process_block(xs[:, B:], B)
    pcm = np.concatenate([r1[0], r2[0]], axis=1)
    mag = np.concatenate([r1[2], r2[2]], axis=1)
    for c in range(C):
        want = _oracle_stream(oracle, mode, xs[c], 2 * B)
        for b in range(2 * B):
            assert (pcm[c, b] == want[b][0]).all(), (mode, kind, c, b)
            assert mag[c, b] == want[b][1]
    assert rx.debug_counters()[5] == 0


@pytest.mark.parametrize("mode", [AM, FM, LSB, USB])
def test_fir_modes_on_the_flow_kernel(oracle, mode):
    """k_rx_wbfm_flow<.., 2> / <.., 14> in depth: block sizes that are whole units of 512 samples at 256 kS/s down to
    the shortest, 17 blocks (more than four generations, ragged last one), three calls (the carried input tail, the FM
    pipelines with the gain of their time, the SSB rails and the dc-removal filter's x[n-1], y[n-1] across calls), a
    gain change between calls, a channel count that is not a multiple of 8, mean magnitudes and n_pcm."""
    for bb, B in ((262144, 3), (65536, 17), (32768, 5), (32768, 45)):
        C = 5
        raw = np.concatenate([synth.make_input("amtone" if c % 2 else "lcg", 210 + c, (3 * B * bb + BLK - 1) // BLK)[: 3 * B * bb]
                              for c in range(C)]).reshape(C, 3 * B, bb)
        rx = api.Rx(C)
        rx.set_mode(mode)
        rx.debug_set_fir_flow(1)
        outs = []
        for call in range(3):
            if call == 2:
                rx.set_gain(mode, 4321.0)
            outs.append(rx.process_block(raw[:, call * B:(call + 1) * B], B))
        for c in range(C):
            o = oracle.rx()
            o.set_mode(mode)
            for call in range(3):
                if call == 2:
                    o.set_gain(mode, 4321.0)
                for b in range(B):
                    p, m, _, _ = o.process(raw[c, call * B + b])
                    got = outs[call]
                    assert got[1][c, b] == len(p) and int(got[2][c, b]) == m, (bb, c, call, b)
                    assert (got[0][c, b, :len(p)] == p).all(), (bb, c, call, b)
        assert rx.debug_counters()[5] == 0, "a launch was not committed"


@pytest.mark.parametrize("fir_flow", [-1, 1, 0], ids=["kernels_per_mode", "one_bank_kernel", "block_kernels"])
def test_mixed_mode_bank(oracle, fir_flow):
    """BASELINE config 3 in miniature: AM + FM + WBFM + LSB + USB + NONE channels in
    one handle, two calls.  A bank this small is dispatched per mode by default; the hook forces what a bank of 48
    channels or more gets: ONE launch of k_rx_flow_bank, one persistent workgroup per channel, the mode read per
    workgroup (mode NONE keeps its own kernel) -- or, with 0, the block kernels of the FIR modes in front of the flow kernel."""
    _mixed_bank(oracle, fir_flow, 3)


def test_mixed_bank_of_long_batches(oracle):
    """the same bank with 21 blocks per call: ONE launch of k_rx_flow_bank as well since round 5 (until then a mixed batch of
    more than 16 blocks went out as a launch per kind), every channel one run of 21 blocks"""
    _mixed_bank(oracle, 1, 21)


def _mixed_bank(oracle, fir_flow, B):
    modes = [AM, FM, WBFM, LSB, USB, NONE, WBFM, AM, FM, USB]
    C = len(modes)
    xs = np.stack([synth.make_input("lcg" if c % 2 else "amtone", 60 + c, 2 * B) for c in range(C)])
    xs = xs.reshape(C, 2 * B, BLK)
    rx = api.Rx(C)
    rx.debug_set_fir_flow(fir_flow)
    for c, m in enumerate(modes):
        rx.set_mode(m, channel=c)
    r1 = rx.process_block(xs[:, :B], B)
    r2 = rx.process_block(xs[:, B:], B, want_iq256=True)  # (the second call with the iq dump: k_rx_flow_bank<DUMP> / <DUMP, MODE>)
    pcm = np.concatenate([r1[0], r2[0]], axis=1)
    npcm = np.concatenate([r1[1], r2[1]], axis=1)
    for c, m in enumerate(modes):
        want = _oracle_stream(oracle, m, xs[c], 2 * B)
        for b in range(2 * B):
            assert npcm[c, b] == len(want[b][0])
            assert (pcm[c, b, :npcm[c, b]] == want[b][0]).all(), (c, m, b)
            if b >= B:
                assert (r2[4][c, b - B] == want[b][3]).all(), (c, m, b)      # the 256 kS/s stream of every channel, whatever its mode


@pytest.mark.parametrize("mode", [AM, FM, WBFM, LSB])
def test_gain_change_between_calls(oracle, mode):
    """the decimator pipelines keep samples scaled with the OLD gain (FM/WBFM)"""
    x = synth.make_input("lcg", 8, 4).reshape(1, 4, BLK)
    rx = api.Rx(1); rx.set_mode(mode)
    o = oracle.rx(); o.set_mode(mode)
    for k, gain in enumerate([None, 777.0, None, 12345.0]):
        if gain is not None:
            rx.set_gain(mode, gain); o.set_gain(mode, gain)
        got = rx.process_block(x[:, k:k + 1], 1)[0][0, 0]
        assert (got == o.process(x[0, k])[0]).all(), (mode, k)


@pytest.mark.parametrize("fir_flow", [1, 0], ids=["one_bank_kernel", "block_kernels"])
def test_mixed_bank_with_closed_gates(oracle, fir_flow):
    """a bank of several modes under a real squelch threshold: gates that close inside the batch in a WBFM channel (the
    device's gated pass redoes it), in FM, AM and SSB channels (their verdict fails; the blocking entry replays them on
    the exact path), a channel that never opens, one that is always open -- every channel the oracle's, two calls"""
    modes = [WBFM, AM, FM, LSB, WBFM, USB, AM, FM]
    pats = ["110011", "101100", "011010", "110001", "111111", "000000", "111111", "100110"]
    C, B = len(modes), len(pats[0])
    xs = np.stack([synth.make_input("fmtone", 90 + c, 2 * B).reshape(2 * B, BLK) for c in range(C)])
    for c, pat in enumerate(pats):
        for b, ch in enumerate(pat + pat[::-1]):
            if ch == "0":
                xs[c, b] = 0
    rx = api.Rx(C)
    rx.debug_set_fir_flow(fir_flow)
    for c, m in enumerate(modes):
        rx.set_mode(m, channel=c)
    rx.set_threshold(-30)
    got = [rx.process_block(xs[:, :B], B), rx.process_block(xs[:, B:], B)]
    for c, m in enumerate(modes):
        want = _oracle_stream(oracle, m, xs[c], 2 * B, threshold=-30)
        for b in range(2 * B):
            r = got[b // B]
            p, mg, a, _ = want[b]
            assert r[1][c, b % B] == len(p) and bool(r[3][c, b % B]) == a and int(r[2][c, b % B]) == mg, (c, m, b)
            assert (r[0][c, b % B, :len(p)] == p).all() and (r[0][c, b % B, len(p):] == 0).all(), (c, m, b)


def test_reduce_sample_rate_advances_the_front_end_only(oracle):
    """IqDataProcessor::reduceSampleRate (IqDataProcessor.cc:429-500) as a call of its own: the 256 kS/s stream of the
    block comes back, the half-band pipelines advance, and neither the squelch tracker nor a demodulator sees the block.
    Sequence: a loud block (the tracker goes to Tracking), a SILENT block through reduceSampleRate only (a squelch run
    would drop the tracker), another silent block through acceptIqData: it must still pass as the tracker's tail block."""
    loud = synth.make_input("fmtone", 3, 1).reshape(BLK)
    x = synth.make_input("lcg", 4, 1).reshape(BLK) // 64         # noise of +-1: below -30 dBFS
    y = synth.make_input("lcg", 5, 1).reshape(BLK) // 64
    rx = api.Rx(1)
    rx.set_mode(api.WBFM)
    rx.set_threshold(-30)
    o = oracle.rx()
    o.set_mode(WBFM)
    o.set_threshold(-30)
    a = rx.process_block(loud.reshape(1, 1, BLK), 1)
    wa = o.process(loud)
    assert (a[0][0, 0] == wa[0]).all()
    got = rx.reduce_sample_rate(x.reshape(1, BLK))
    # the oracle has no such entry: the same effect is a mode-NONE block whose squelch leaves the tracker alone
    o.set_mode(NONE)
    o.set_threshold(-200)
    wx = o.process(x)
    assert (got[0] == wx[3]).all()
    o.set_mode(WBFM)
    o.set_threshold(-30)
    b = rx.process_block(y.reshape(1, 1, BLK), 1)
    wb = o.process(y)
    assert wb[2] and len(wb[0]) == 512, "the oracle passes the tail block"
    assert b[1][0, 0] == 512 and bool(b[3][0, 0]) and (b[0][0, 0] == wb[0]).all()


def test_mode_switch_keeps_each_demodulators_state(oracle):
    x = synth.make_input("lcg", 5, 6).reshape(1, 6, BLK)
    rx = api.Rx(1)
    o = oracle.rx()
    for blk, mode in enumerate([WBFM, AM, LSB, USB, FM, WBFM]):
        rx.set_mode(mode)
        o.set_mode(mode)
        got = rx.process_block(x[:, blk:blk + 1], 1)[0][0, 0]
        assert (got == o.process(x[0, blk])[0]).all(), (blk, mode)


@pytest.mark.parametrize("kind", ["lcg", "fmtone", "dc_pos", "dc_neg", "impulse", "zeros"])
def test_batched_blocks_match_oracle(oracle, kind):
    """blocks of one channel demodulated concurrently (time-parallel) must equal
    the strictly sequential reference, and the state must carry into the next call."""
    C, B = 5, 4
    xs = np.stack([synth.make_input(kind, 20 + c, 2 * B) for c in range(C)]).reshape(C, 2 * B, BLK)
    rx = api.Rx(C)
    rx.set_mode(api.WBFM)
    r1 = rx.process_block(xs[:, :B], B, want_iq256=True)
    r2 = rx.process_block(xs[:, B:], B, want_iq256=True)
    pcm = np.concatenate([r1[0], r2[0]], axis=1)
    mag = np.concatenate([r1[2], r2[2]], axis=1)
    iq256 = np.concatenate([r1[4], r2[4]], axis=1)
    assert (r1[1] == 512).all() and (r2[1] == 512).all()
    for c in range(C):
        want = _oracle_stream(oracle, WBFM, xs[c], 2 * B)
        for b in range(2 * B):
            assert (pcm[c, b] == want[b][0]).all(), (c, b)
            assert mag[c, b] == want[b][1]
            assert (iq256[c, b] == want[b][3]).all()
    cnt = rx.debug_counters()
    assert cnt[5] == 0 and cnt[7] == 0          # nothing uncommitted, nothing replayed


def test_mode_none_produces_magnitude_only(oracle):
    x = synth.make_input("fmtone", 9, 2).reshape(1, 2, BLK)
    rx = api.Rx(1)
    pcm, n_pcm, mag, allowed, iq256 = rx.process_block(x, 2, want_iq256=True)
    want = _oracle_stream(oracle, NONE, x[0], 2)
    assert (n_pcm == 0).all() and (allowed == 1).all()
    for b in range(2):
        assert mag[0, b] == want[b][1] and (iq256[0, b] == want[b][3]).all()


def test_mixed_none_and_wbfm_channels(oracle):
    C, B = 6, 2
    xs = np.stack([synth.make_input("lcg", 50 + c, B) for c in range(C)]).reshape(C, B, BLK)
    rx = api.Rx(C)
    for c in range(C):
        rx.set_mode(api.WBFM if c % 2 else api.NONE, channel=c)
    pcm, n_pcm, mag, _, _ = rx.process_block(xs, B)
    for c in range(C):
        want = _oracle_stream(oracle, WBFM if c % 2 else NONE, xs[c], B)
        for b in range(B):
            assert n_pcm[c, b] == len(want[b][0]) and mag[c, b] == want[b][1]
            assert (pcm[c, b, :n_pcm[c, b]] == want[b][0]).all()


@pytest.mark.parametrize("mode", [AM, FM, WBFM, LSB])
@pytest.mark.parametrize("gain", [1.0, 1234.5, 1e6, 1e12])
def test_gain_incl_float_to_int16_wrap(oracle, mode, gain):
    x = synth.make_input("lcg", 11, 2).reshape(1, 2, BLK)
    rx = api.Rx(1)
    rx.set_mode(mode)
    rx.set_gain(mode, gain)
    pcm = rx.process_block(x, 2)[0]
    want = _oracle_stream(oracle, mode, x[0], 2, gain=gain)
    for b in range(2):
        assert (pcm[0, b] == want[b][0]).all()


@pytest.mark.parametrize("mode", [AM, FM, WBFM, USB])
@pytest.mark.parametrize("bb", [1024, 4096, 32768, 65536, 262144])
def test_block_sizes(oracle, mode, bb):
    nb = 6
    x = synth.make_input("fmtone", 4, 6)[: nb * bb].reshape(1, nb, bb)
    rx = api.Rx(1)
    rx.set_mode(mode)
    a = rx.process_block(x[:, :3], 3)[0]
    b = rx.process_block(x[:, 3:], 3)[0]
    got = np.concatenate([a, b], axis=1).reshape(-1)
    o = oracle.rx(); o.set_mode(mode)
    want = np.concatenate([o.process(x[0, k])[0] for k in range(nb)])
    assert (got == want).all()


def test_invalid_sizes_are_rejected():
    """what the reference itself cannot take (any other even length is demodulated: tests/test_gpu_short_blocks.py)"""
    rx = api.Rx(1)
    with pytest.raises(api.HrfdError):
        rx.process_block(np.zeros((1, 1, 1001), dtype=np.int8), 1)     # odd: the reference's Q loop reads past the end (IqDataProcessor.cc:474)
    with pytest.raises(api.HrfdError):
        rx.process_block(np.zeros((1, 1, 2 * 262144), dtype=np.int8), 1)  # larger than its arrays (DataConsumer clips first, DataConsumer.cc:229-233)


def test_squelch_in_a_batch_falls_back_to_exact_path(oracle):
    """a closed gate inside a multi-block call breaks the 'all gates open'
    speculation: the call must notice, not commit, and replay block by block."""
    loud = synth.make_input("fmtone", 1, 1)
    quiet = synth.zeros_iq(synth.BLOCK_IQ)
    pattern = [1, 1, 0, 0, 1, 0, 0, 0, 1]
    x = np.stack([loud if p else quiet for p in pattern]).reshape(1, len(pattern), BLK)
    rx = api.Rx(1)
    rx.set_mode(api.WBFM)
    rx.set_threshold(-30)
    pcm, n_pcm, mag, allowed, _ = rx.process_block(x, len(pattern))
    want = _oracle_stream(oracle, WBFM, x[0], len(pattern), threshold=-30)
    for b in range(len(pattern)):
        assert n_pcm[0, b] == len(want[b][0]) and bool(allowed[0, b]) == want[b][2]
        assert (pcm[0, b, :n_pcm[0, b]] == want[b][0]).all()
    assert rx.debug_counters()[7] == 0          # replays only count de-emphasis redo, none needed


@pytest.mark.parametrize("run_len", [0, 16], ids=["several_runs_per_channel", "one_workgroup_per_channel"])
def test_one_quiet_channel_does_not_stop_the_bank(oracle, run_len):
    """(run_len 16: every channel is ONE workgroup's, which finishes it from its LDS -- the path of a full bank;
    run_len 0, automatic: a small bank is cut into several runs per channel, finished by the workgroup that arrives last.)
    The verdict of a batch is per channel: with a real squelch threshold, one channel whose gate closes
    inside the batch fails ITS speculation and is replayed block by block; the other channels commit from
    the batch launch (device path: sync() reports one failed channel and says which).  Squelched units
    hand back zeros, not what the failed batch launch left in the buffer."""
    import torch
    C, B = 6, 5
    loud = [synth.make_input("fmtone", 40 + c, B).reshape(B, BLK) for c in range(C)]
    xs = np.stack(loud)
    quiet_c = 3
    xs[quiet_c, 1:4] = 0                                  # the gate of channel 3 closes for blocks 2 and 3 (one tail block)
    # blocking entry: everything exact, squelched PCM zero
    rx = api.Rx(C)
    rx.set_mode(api.WBFM)
    rx.set_threshold(-30)
    rx.debug_set_run_len(run_len)
    pcm, n_pcm, mag, allowed, _ = rx.process_block(xs, B)
    for c in range(C):
        want = _oracle_stream(oracle, WBFM, xs[c], B, threshold=-30)
        for b in range(B):
            assert n_pcm[c, b] == len(want[b][0]) and bool(allowed[c, b]) == want[b][2], (c, b)
            assert (pcm[c, b, :n_pcm[c, b]] == want[b][0]).all(), (c, b)
            assert (pcm[c, b, n_pcm[c, b]:] == 0).all(), (c, b)
    assert (n_pcm[quiet_c] == 0).sum() == 2
    # device entry without the gated second pass (test hook): one launch, per-channel verdict
    dev = torch.device("cuda:0")
    rx2 = api.Rx(C)
    rx2.set_mode(api.WBFM)
    rx2.set_threshold(-30)
    rx2.debug_set_run_len(run_len)
    rx2.debug_set_gated(False)
    x = torch.from_numpy(xs).to(dev)
    out = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
    torch.cuda.synchronize()
    rx2.process_device(x.data_ptr(), B * BLK, BLK, B, out.data_ptr())
    assert rx2.sync() == 1
    assert rx2.failed_channels().tolist() == [1 if c == quiet_c else 0 for c in range(C)]
    got = out.cpu().numpy()
    for c in range(C):
        if c != quiet_c:
            assert (got[c] == pcm[c]).all(), c             # committed from the batch, exact
    # the clean channels have advanced: a second batch continues them; the failed one has not
    xs2 = np.stack([synth.make_input("fmtone", 40 + c, 2 * B).reshape(2 * B, BLK)[B:] for c in range(C)])
    x2 = torch.from_numpy(xs2).to(dev)
    torch.cuda.synchronize()
    rx2.process_device(x2.data_ptr(), B * BLK, BLK, B, out.data_ptr())
    rx2.sync()
    got2 = out.cpu().numpy()
    pcm2 = rx.process_block(xs2, B)[0]
    for c in range(C):
        if c != quiet_c:
            assert (got2[c] == pcm2[c]).all(), c


@pytest.mark.parametrize("mode,run_len", [(WBFM, 0), (WBFM, 16), (AM, 0), (FM, 0), (LSB, 0)],
                         ids=["wbfm_several_runs_per_channel", "wbfm_one_workgroup_per_channel", "am", "fm", "lsb"])
def test_closed_gates_are_redone_on_the_device(oracle, mode, run_len):
    """Squelch inside a batch without the host (IqDataProcessor.cc:961-1034, Squelch.cc:227-273, SignalTracker.cc:104-146):
    the batch launch speculates every gate open; the gated pass behind it (k_rx_wbfm_flow<GATED>, every mode) redoes, on
    the device, the channels whose gates closed -- the stream of the ALLOWED blocks only, from the committed state,
    squelched blocks zero -- so hrfd_rx_sync reports no failed channel and every output is the oracle's: PCM, n_pcm,
    signal_allowed, magnitude, and the state a second batch continues from (the tracker's tail block included).
    Gate patterns: never open, closing and reopening, open only at the end, the tracker's tail across the call
    boundary, always open."""
    import torch
    patterns = ["0000000", "1100110", "0000011", "1000000", "0101010", "1111111", "0011100", "1110001"]
    C, B = len(patterns), len(patterns[0])
    loud = [synth.make_input("fmtone", 80 + c, 2 * B).reshape(2 * B, BLK) for c in range(C)]
    xs = np.stack(loud)
    for c, pat in enumerate(patterns):
        for b, ch in enumerate(pat):
            if ch == "0":
                xs[c, b] = 0 if (b + c) % 2 else synth.make_input("lcg", 5, 1).reshape(BLK) // 64   # silence, or noise of +-1
        for b, ch in enumerate(pat[::-1]):                 # the second batch: the pattern backwards
            if ch == "0":
                xs[c, B + b] = 0
    want = [_oracle_stream(oracle, mode, xs[c], 2 * B, threshold=-30) for c in range(C)]
    dev = torch.device("cuda:0")
    rx = api.Rx(C)
    rx.set_mode(mode)
    rx.set_threshold(-30)
    rx.debug_set_run_len(run_len)
    rx.debug_set_fir_flow(1)                                # (AM / FM / SSB: the flow kernel's FIR modes, whatever the bank's size)
    out = torch.full((C, B, 512), 777, dtype=torch.int16, device=dev)
    npcm = torch.zeros((C, B), dtype=torch.int32, device=dev)
    mag = torch.zeros((C, B), dtype=torch.int32, device=dev)
    alw = torch.zeros((C, B), dtype=torch.uint8, device=dev)
    for half in range(2):
        x = torch.from_numpy(xs[:, half * B:(half + 1) * B].copy()).to(dev)
        out.fill_(777)
        torch.cuda.synchronize()
        rx.process_device(x.data_ptr(), B * BLK, BLK, B, out.data_ptr(), d_n_pcm=npcm.data_ptr(), d_magnitude=mag.data_ptr(),
                          d_allowed=alw.data_ptr())
        assert rx.sync() == 0, rx.failed_channels()
        got, gn, gm, ga = out.cpu().numpy(), npcm.cpu().numpy(), mag.cpu().numpy(), alw.cpu().numpy()
        for c in range(C):
            for b in range(B):
                p, m, a, _ = want[c][half * B + b]
                assert gn[c, b] == len(p) and bool(ga[c, b]) == a and gm[c, b] == m, (half, c, b)
                if len(p):
                    assert (got[c, b] == p).all(), (half, c, b)
                elif patterns[c] != "1111111":
                    assert (got[c, b] == 0).all(), (half, c, b)      # a channel the gated pass redid: silence
    assert any(len(want[c][b][0]) == 0 for c in range(C) for b in range(2 * B))


def test_long_call_with_closing_gates_is_repaired_on_the_device_chunk_by_chunk(oracle):
    """Round 6 ("+6.4 ms host replay"): hrfd_rx_process_block runs a call of more than 64 blocks as chunks of
    at most 64, each a batch launch with the gated pass behind it, so closing gates in a long call are repaired on the
    device like in a short one.  150 blocks of 32 KiB per channel (chunks of 64, 64 and 22), every mode in the bank, gates
    that close and reopen inside chunks AND across the chunk boundaries (the tracker's tail block is block 64 / block 128
    of a channel), one channel that never opens, one that never closes; a second call continues.  Everything = the
    sequential oracle, and the host replayed nothing."""
    bb, B = 32768, 150
    modes = [WBFM, AM, FM, LSB, USB, WBFM, WBFM, AM]
    C = len(modes)
    quiet = {0: [(10, 20), (60, 70), (120, 131)], 1: [(63, 64)], 2: [(64, 65), (127, 129)], 3: [(0, 66)], 4: [(62, 128)],
             5: [(0, B)], 6: [], 7: [(30, 31), (33, 34), (100, 149)]}
    need = (2 * B * bb + BLK - 1) // BLK
    xs = np.stack([synth.make_input("fmtone", 300 + c, need)[:2 * B * bb].reshape(2 * B, bb) for c in range(C)])
    for c, spans in quiet.items():
        for a, b in spans:
            xs[c, a:b] = 0
            xs[c, B + a:B + b] = 0 if c % 2 else xs[c, B + a:B + b] // 64
    rx = api.Rx(C)
    orc = []
    for c in range(C):
        rx.set_mode(modes[c], channel=c)
        o = oracle.rx(); o.set_mode(modes[c]); o.set_threshold(-30); orc.append(o)
    rx.set_threshold(-30)
    from tests.hooks import HOOKS_ON
    if HOOKS_ON:
        # a bank of eight gets the shapes of a full one: one run per channel and chunk, the FIR modes on the flow kernel
        # (a small bank runs them on the block kernels, which leave closed gates to the host: still exact, not what is asked here)
        rx.debug_set_run_len(64)
        rx.debug_set_fir_flow(1)
    squelched = 0
    for call in range(2):
        x = xs[:, call * B:(call + 1) * B]
        pcm, n_pcm, mag, allowed, _ = rx.process_block(x, B)
        for c in range(C):
            for b in range(B):
                p, m, a, _ = orc[c].process(x[c, b])
                assert n_pcm[c, b] == len(p) and int(mag[c, b]) == m and bool(allowed[c, b]) == a, (call, c, b)
                assert (pcm[c, b, :len(p)] == p).all(), (call, c, b)
                squelched += len(p) == 0
    assert squelched > 300
    if HOOKS_ON:
        assert rx.debug_counters()[5] == 0, "a chunk left a channel uncommitted (the host replayed it block by block)"


@pytest.mark.parametrize("mode,point", [(AM, 1), (AM, 2), (LSB, 1), (LSB, 2), (LSB, 3), (FM, 1), (WBFM, 4), (WBFM, 5), (WBFM, 6),
                                        (WBFM, 7), (AM, 7), (FM, 7), (WBFM, 8), (WBFM, 9), (WBFM, 10)],
                         ids=["am_b", "am_c", "lsb_b", "lsb_c", "lsb_rails", "fm_b", "wbfm_sums", "wbfm_verify", "wbfm_integer",
                              "wbfm_stream", "am_stream", "fm_stream", "wbfm_arrival", "wbfm_rows_released", "wbfm_pass_carry"])
@pytest.mark.parametrize("gen", [0, 1, 6, 20])
def test_a_held_up_service_wave_is_not_overtaken(oracle, mode, point, gen):
    """The service waves of the flow kernel hand generations over to each other at a few points and otherwise run side by
    side, over rings that hold a few generations (v: six; AM / SSB: four of the second decimator's output and of SSB's
    8 kS/s rails).  hrfd_rx_debug_expire(1000 p + g) holds the wave of generation g of workgroup 0 up for ~60 us right
    behind hand-over point p, while the generations behind it run on: none of them may write over what the held-up one
    (and the one behind it) still have to read.  Also held up: a stream wave behind publishing a unit (point 7), and a
    workgroup in front of its arrival at its channel's count (point 8; the channels are cut into runs here, one block per
    workgroup, so that another workgroup finishes the channel).  PCM = the oracle's, nothing reported as failed.  (Found as one SSB
    channel of a mixed bank with wrong PCM in one launch of many: AM / SSB let the four generations behind a wave that
    was slow between its part c and its 8 kS/s part write into its ring -- point 2 fails without the wait that was
    added for it.)"""
    import torch
    C, B = 2, 8
    xs = np.stack([synth.make_input("fmtone" if c else "lcg", 70 + c, B).reshape(B, BLK) for c in range(C)])
    want = [_oracle_stream(oracle, mode, xs[c], B) for c in range(C)]
    dev = torch.device("cuda:0")
    rx = api.Rx(C)
    rx.set_mode(mode)
    rx.debug_set_fir_flow(1)
    x = torch.from_numpy(xs).to(dev)
    out = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
    torch.cuda.synchronize()
    rx.debug_expire(1000 * point + gen)
    rx.process_device(x.data_ptr(), B * BLK, BLK, B, out.data_ptr())
    assert rx.sync() == 0, rx.failed_channels()
    got = out.cpu().numpy()
    for c in range(C):
        for b in range(B):
            assert (got[c, b] == want[c][b][0]).all(), (c, b)


def test_expired_block_slot_wait_is_replayed(oracle):
    """wait 15 of the flow kernel -- a unit of block b >= 16 waits for block b - 16 to be finished before it
    takes its magnitude slot (runs of more than 16 blocks: since round 5 up to 64) -- forced to expire: 40 blocks of the
    shortest size (32 KiB: a stream wave reaches block 16 while block 0's sum is still out), one run per channel.  The
    channel concerned is reported and not committed (or the wait was never polled: then everything committed), the
    blocking entry replays it, and the PCM is the oracle's either way, over two calls."""
    bb, B, C = 32768, 40, 3
    need = (2 * B * bb + BLK - 1) // BLK
    xs = np.stack([synth.make_input("fmtone", 700 + c, need)[:2 * B * bb].reshape(2 * B, bb) for c in range(C)])
    rx = api.Rx(C)
    rx.set_mode(api.WBFM)
    rx.debug_set_run_len(64)
    orc = []
    for c in range(C):
        o = oracle.rx(); o.set_mode(WBFM); orc.append(o)
    for call in range(2):
        rx.debug_expire(15)
        x = xs[:, call * B:(call + 1) * B]
        pcm, n_pcm, mag, allowed, _ = rx.process_block(x, B)
        for c in range(C):
            for b in range(B):
                p, m, a, _ = orc[c].process(x[c, b])
                assert n_pcm[c, b] == len(p) and int(mag[c, b]) == m, (call, c, b)
                assert (pcm[c, b, :len(p)] == p).all(), (call, c, b)


@pytest.mark.parametrize("run_len", [0, 16], ids=["several_runs_per_channel", "one_workgroup_per_channel"])
@pytest.mark.parametrize("where", [3, 5, 6, 1])
def test_expired_wait_fails_the_channel_and_is_replayed(oracle, where, run_len):
    """The bounded-spin failure path of k_rx_wbfm_flow (FlowSpin: kFailExpired, the workgroup's abort word, host
    replay).  hrfd_rx_debug_expire makes workgroup 0 of the next launch treat one of its waits as expired the first
    time it polls it: the workgroup must drain at once (not one spin limit per wait), its channel must be reported
    as failed with bit 8 and must not commit, the other channels commit, and the blocking entry -- which replays
    the failed channel on the exact path -- still returns the oracle's PCM for every channel.  Wait 3 (a
    generation's units) is polled by every workgroup at its start, so its failure is certain; the others (1 ring
    space, 5 / 6 generation order) are polled only when a wave actually has to wait: either way the results are exact."""
    import time
    import torch
    C, B = 3, 4
    xs = np.stack([synth.make_input("fmtone", 60 + c, B).reshape(B, BLK) for c in range(C)])
    want = [_oracle_stream(oracle, WBFM, xs[c], B) for c in range(C)]
    dev = torch.device("cuda:0")
    rx = api.Rx(C)
    rx.set_mode(api.WBFM)
    rx.debug_set_run_len(run_len)
    x = torch.from_numpy(xs).to(dev)
    out = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
    torch.cuda.synchronize()
    rx.debug_expire(where)
    t0 = time.perf_counter()
    rx.process_device(x.data_ptr(), B * BLK, BLK, B, out.data_ptr())
    failed = rx.sync()
    assert time.perf_counter() - t0 < 2.0, "an aborting workgroup must drain promptly"
    f = rx.failed_channels()
    if where == 3:
        assert failed == 1 and f[0] & 8, (failed, f)
    assert failed == int((f != 0).sum()) and failed <= 1 and not f[1:].any()
    got = out.cpu().numpy()
    for c in range(C):
        if f[c] == 0:
            assert all((got[c, b] == want[c][b][0]).all() for b in range(B)), c
    # the blocking entry repairs the channel by itself
    rx2 = api.Rx(C)
    rx2.set_mode(api.WBFM)
    rx2.debug_set_run_len(run_len)
    rx2.debug_expire(where)
    pcm = rx2.process_block(xs, B)[0]
    for c in range(C):
        assert all((pcm[c, b] == want[c][b][0]).all() for b in range(B)), c
    # ... and the streams go on from the right state
    xs2 = np.stack([synth.make_input("fmtone", 60 + c, 2 * B).reshape(2 * B, BLK)[B:] for c in range(C)])
    pcm2 = rx2.process_block(xs2, B)[0]
    for c in range(C):
        w2 = _oracle_stream(oracle, WBFM, np.concatenate([xs[c], xs2[c]]), 2 * B)
        assert all((pcm2[c, b] == w2[B + b][0]).all() for b in range(B)), c


@pytest.mark.parametrize("warm", [64, 256, 384])
def test_short_warmup_is_repaired_exactly(oracle, warm):
    """Shrinking the de-emphasis warm-up makes tiles fail to re-synchronise; the
    verify-and-repair step must still deliver the sequential result bit for bit."""
    C, B = 3, 3
    xs = np.stack([synth.make_input("fmtone" if c else "lcg", 70 + c, B) for c in range(C)]).reshape(C, B, BLK)
    rx = api.Rx(C)
    rx.set_mode(api.WBFM)
    rx.debug_set_warm(warm)
    pcm = rx.process_block(xs, B)[0]
    for c in range(C):
        want = _oracle_stream(oracle, WBFM, xs[c], B)
        for b in range(B):
            assert (pcm[c, b] == want[b][0]).all(), (warm, c, b)
    assert rx.debug_counters()[4] > 0           # the repair path really ran


def test_reset_demod_wbfm(oracle):
    x = synth.lcg_bytes(21, 3 * BLK).reshape(1, 3, BLK)
    rx = api.Rx(1)
    rx.set_mode(api.WBFM)
    o = oracle.rx(); o.set_mode(WBFM)
    for k in range(3):
        if k == 2:
            rx.reset_demod(api.WBFM)
            o.lib.orc_demod_reset  # (the outer oracle object has no reset; emulate via inner API below)
        got = rx.process_block(x[:, k:k + 1], 1)[0][0, 0]
        if k < 2:
            assert (got == o.process(x[0, k])[0]).all()
    # after a reset the decimator pipelines and previousTheta are zero but the
    # de-emphasis filter keeps its state (WbFmDemodulator.cc:265-278): compare with
    # the inner-API oracle driven by the same 256 kS/s stream.
    d = oracle.demod(WBFM)
    f = oracle.rx()
    iq256 = [f.process(x[0, k])[3] for k in range(3)]
    want = []
    for k in range(3):
        if k == 2:
            d.reset()
        want.append(d.process(iq256[k]))
    assert (got == want[2]).all()


def test_device_entry_and_sync(oracle):
    import torch
    C, B = 4, 3
    xs = np.stack([synth.make_input("lcg", 90 + c, B) for c in range(C)]).reshape(C, B, BLK)
    dev = torch.device("cuda:0")
    d_iq = torch.from_numpy(xs).to(dev)
    d_pcm = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
    d_np = torch.zeros((C, B), dtype=torch.int32, device=dev)
    d_mag = torch.zeros((C, B), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()                             # torch filled these on its own stream
    rx = api.Rx(C)
    rx.set_mode(api.WBFM)
    s = torch.cuda.Stream()
    rx.process_device(d_iq.data_ptr(), B * BLK, BLK, B, d_pcm.data_ptr(), d_n_pcm=d_np.data_ptr(),
                      d_magnitude=d_mag.data_ptr(), stream=s.cuda_stream)
    assert rx.sync() == 0
    pcm = d_pcm.cpu().numpy()
    for c in range(C):
        want = _oracle_stream(oracle, WBFM, xs[c], B)
        for b in range(B):
            assert (pcm[c, b] == want[b][0]).all() and int(d_mag[c, b]) == want[b][1]
    assert (d_np.cpu().numpy() == 512).all()


# ---------------------------------------------------------------- inner boundary (per-demodulator API)
@pytest.mark.parametrize("mode", [AM, FM, WBFM, LSB, USB])
def test_inner_demod_api_matches_oracle(oracle, mode):
    """X::acceptIqData on the 256 kS/s mixed stream, 32768-byte calls as IqDataProcessor
    makes them, incl. resetDemodulator and setDemodulatorGain between calls."""
    x = synth.lcg_bytes(21, 5 * 32768)
    g, o = api.Demod(mode, 1), oracle.demod(mode)
    for k in range(5):
        if k == 2:
            g.reset(); o.reset()
        if k == 3:
            g.set_gain(777.0); o.set_gain(777.0)
        a = g.process(x[k * 32768:(k + 1) * 32768])
        b = o.process(x[k * 32768:(k + 1) * 32768])
        assert len(a) == 512 and (a == b).all(), (mode, k)


def test_inner_demod_sideband_switch_and_sizes(oracle):
    x = synth.lcg_bytes(3, 4 * 32768)
    g, o = api.Demod(LSB, 1), oracle.demod(LSB)
    off = 0
    for n, lsb in [(32768, True), (4096, False), (128, False), (16384, True)]:
        g.set_sideband(lsb); o.set_sideband(lsb)
        a = g.process(x[off:off + n]); b = o.process(x[off:off + n])
        assert (a == b).all(), n
        off += n
    # round 6: any even count, like the reference's loops (WbFmDemodulator.cc:395): short and off-grid pieces continue the stream
    for n in (64, 2, 190, 4096):
        a = g.process(x[off:off + n]); b = o.process(x[off:off + n])
        assert len(a) == len(b) and (a == b).all(), n
        off += n
    with pytest.raises(api.HrfdError):
        g.process(np.zeros(63, dtype=np.int8))          # an odd count is half an IQ pair
    with pytest.raises(api.HrfdError):
        g.process(np.zeros(32770, dtype=np.int8))       # the reference's member arrays hold 32768
    with pytest.raises(api.HrfdError):
        api.Demod(WBFM, 1).set_sideband(True)


def test_inner_demod_many_channels(oracle):
    C = 7
    xs = np.stack([synth.lcg_bytes(40 + c, 2 * 32768) for c in range(C)])
    g = api.Demod(WBFM, C)
    got = [g.process(xs[:, k * 32768:(k + 1) * 32768]) for k in range(2)]
    for c in range(C):
        o = oracle.demod(WBFM)
        for k in range(2):
            assert (got[k][c] == o.process(xs[c, k * 32768:(k + 1) * 32768])).all()


# ---------------------------------------------------------------- arithmetic atan2
@pytest.mark.parametrize("tab", [False, True, "quad"], ids=["polynomial", "first_octant_table", "first_quadrant_table"])
def test_arithmetic_atan2_equals_table_everywhere(tab):
    """The WBFM kernels compute theta instead of gathering it from the reference's 256 x 256
    table -- k_rx_wbfm: polynomial + 2-bit correction from LDS; k_rx_wbfm_flow:
    first-octant float table (8385 entries) + octant arithmetic + 2-bit correction; the re-split WBFM flow kernel
    (round 5): first-QUADRANT table (16641 words, the correction of the i < 0 half in their two free top bits).
    Every one of the 65536 (q, i) entries must be the table's float, bit for bit."""
    rx = api.Rx(1)
    rx.debug_set_atan(1)                         # raises if the corrections did not fit
    got = rx.debug_atan_eval(tab)
    want = api.atan2_table()
    assert (got.view(np.uint32) == want.view(np.uint32)).all()


@pytest.mark.parametrize("kind", ["lcg", "fmtone", "dc_neg", "impulse", "fullscale"])
def test_atan2_kernels_agree(oracle, kind):
    """both builds of the WBFM kernel (table gather, arithmetic) against the oracle"""
    C, B = 2, 3
    if kind == "fullscale":
        # every byte at +-127/-128: the decimated samples reach the table's rim
        rng = np.random.default_rng(5)
        xs = rng.choice(np.array([-128, -127, 127], dtype=np.int8), size=(C, B, BLK))
        # long runs so that the half-band stages pass the extremes through
        xs = np.repeat(xs[:, :, ::64], 64, axis=2).copy()
    else:
        xs = np.stack([synth.make_input(kind, 90 + c, B) for c in range(C)]).reshape(C, B, BLK)
    out = []
    for mode in (0, 1):
        rx = api.Rx(C)
        rx.set_mode(api.WBFM)
        rx.debug_set_atan(mode)
        out.append(rx.process_block(xs, B)[0])
    assert (out[0] == out[1]).all()
    for c in range(C):
        want = _oracle_stream(oracle, WBFM, xs[c], B)
        for b in range(B):
            assert (out[1][c, b] == want[b][0]).all(), (c, b)


# ---------------------------------------------------------------- runs of blocks per workgroup
@pytest.mark.parametrize("run_len", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("kind", ["lcg", "fmtone"])
def test_wbfm_block_runs_match_oracle(oracle, run_len, kind):
    """a WBFM workgroup walks `run_len` consecutive blocks of its channel, carrying the tail of
    the phase-difference stream from block to block instead of re-producing it: any run length
    (incl. ones that do not divide the block count) must give the sequential result, over two
    calls so that the carried state is exercised too"""
    C, B = 3, 7
    xs = np.stack([synth.make_input(kind, 50 + c, 2 * B) for c in range(C)]).reshape(C, 2 * B, BLK)
    rx = api.Rx(C)
    rx.set_mode(api.WBFM)
    rx.debug_set_run_len(run_len)
    r1 = rx.process_block(xs[:, :B], B, want_iq256=True)
    r2 = rx.process_block(xs[:, B:], B, want_iq256=True)
    pcm = np.concatenate([r1[0], r2[0]], axis=1)
    mag = np.concatenate([r1[2], r2[2]], axis=1)
    iq256 = np.concatenate([r1[4], r2[4]], axis=1)
    for c in range(C):
        want = _oracle_stream(oracle, WBFM, xs[c], 2 * B)
        for b in range(2 * B):
            assert (pcm[c, b] == want[b][0]).all(), (run_len, c, b)
            assert int(mag[c, b]) == want[b][1], (run_len, c, b)   # the squelch magnitude of every block
            assert (iq256[c, b] == want[b][3]).all(), (run_len, c, b)   # and the 256 kS/s dump stream
    assert rx.debug_counters()[5] == 0           # every launch verified clean and was committed


@pytest.mark.parametrize("run_len", [2, 4])
def test_wbfm_block_runs_with_repairs(oracle, run_len):
    """short warm-up forces tile repairs, also in continuation blocks (whose history comes from
    the carried tail) -- still bit-exact"""
    C, B = 2, 4
    xs = np.stack([synth.make_input("fmtone" if c else "lcg", 80 + c, B) for c in range(C)]).reshape(C, B, BLK)
    rx = api.Rx(C)
    rx.set_mode(api.WBFM)
    rx.debug_set_run_len(run_len)
    rx.debug_set_warm(256)
    pcm = rx.process_block(xs, B)[0]
    for c in range(C):
        want = _oracle_stream(oracle, WBFM, xs[c], B)
        for b in range(B):
            assert (pcm[c, b] == want[b][0]).all(), (run_len, c, b)
    assert rx.debug_counters()[4] > 0


@pytest.mark.parametrize("kind", ["lcg", "fmtone", "fullscale"])
def test_fm_atan2_kernels_agree(oracle, kind):
    """the FM kernel too exists with the table gather and with the arithmetic atan2 (its index is
    the low byte of the int16 tuner outputs): both against the oracle, bit for bit"""
    C, B = 2, 3
    if kind == "fullscale":
        rng = np.random.default_rng(6)
        xs = rng.choice(np.array([-128, -127, 127], dtype=np.int8), size=(C, B, BLK))
        xs = np.repeat(xs[:, :, ::256], 256, axis=2).copy()
    else:
        xs = np.stack([synth.make_input(kind, 95 + c, B) for c in range(C)]).reshape(C, B, BLK)
    out = []
    for mode in (0, 1):
        rx = api.Rx(C)
        rx.set_mode(api.FM)
        rx.debug_set_atan(mode)
        out.append(rx.process_block(xs, B)[0])
    assert (out[0] == out[1]).all()
    for c in range(C):
        want = _oracle_stream(oracle, FM, xs[c], B)
        for b in range(B):
            assert (out[1][c, b] == want[b][0]).all(), (c, b)


def test_full_size_bench_batch_matches_oracle(oracle):
    """BASELINE config 2 at full size -- 256 WBFM channels x 16 blocks in one launch (1 GiB of IQ) followed by a
    second launch that continues the streams, EVERY channel fed an input of its own (the bench's own generators,
    tests/fullsize.py): every PCM sample, magnitude and gate of 64 channels drawn from the whole range (0, 255, every
    residue mod 8) against the sequential oracle, the block kernel agreeing on all 256, nothing replayed.  Then the
    all-channel property with inputs of period 7 (coprime to the 8 XCDs): equal input => equal PCM."""
    from tests.fullsize import check_rx_bank_distinct, check_rx_bank_period
    check_rx_bank_distinct(oracle, api, 256, 16, lambda c: WBFM, twin=lambda rx2: rx2.debug_set_stream(0))
    check_rx_bank_period(oracle, api, 256, 16, lambda c: WBFM, seed=200)


def test_odd_shapes_mixed_modes_and_runs(oracle):
    """37 channels (not a multiple of the 8 XCDs) in five modes, 5 blocks of 65536 bytes per call,
    forced runs of 3 blocks, two calls: every channel against its own sequential oracle"""
    C, B, bb = 37, 5, 65536
    modes = [WBFM, AM, FM, LSB, USB, WBFM, NONE]
    raw = [synth.make_input("fmtone" if c % 3 else "lcg", 300 + c, 3)[: 2 * B * bb].reshape(2 * B, bb) for c in range(C)]
    xs = np.stack(raw)
    rx = api.Rx(C)
    for c in range(C):
        rx.set_mode(modes[c % len(modes)], channel=c)
    rx.debug_set_run_len(3)
    r1 = rx.process_block(xs[:, :B], B)
    r2 = rx.process_block(xs[:, B:], B)
    pcm = np.concatenate([r1[0], r2[0]], axis=1)
    n_pcm = np.concatenate([r1[1], r2[1]], axis=1)
    for c in range(C):
        m = modes[c % len(modes)]
        o = oracle.rx(); o.set_mode(m)
        for b in range(2 * B):
            want = o.process(xs[c, b])[0]
            assert n_pcm[c, b] == len(want), (c, m, b)
            assert (pcm[c, b, :len(want)] == want).all(), (c, m, b)


@pytest.mark.parametrize("stream", [True, False], ids=["flow_kernel_where_it_applies", "per_block_kernel"])
@pytest.mark.parametrize("bb,dump", [(262144, False), (262144, True), (65536, True), (36864, False), (21504, False), (21504, True)],
                         ids=["full", "full_iqdump", "n256_4096_iqdump", "n256_2304", "n256_1344", "n256_1344_iqdump"])
def test_wbfm_batches_on_both_kernels_and_layouts(oracle, stream, bb, dump):
    """WBFM batches through k_rx_wbfm_flow (the default where the block is whole units of 512 samples at 256 kS/s,
    with and without the iq dump: k_rx_wbfm_flow<DUMP>) and through k_rx_wbfm (the hook, and the default for the
    other block sizes): block sizes down to the shortest a batch may have, runs that start with re-derived history
    (run_len 2 and 3) -- PCM, magnitudes and the 256 kS/s dump bit-exact, every launch committed."""
    C, B = 3, 6
    n = C * B * bb
    raw = np.concatenate([synth.make_input("fmtone" if c else "lcg", 120 + c, (B * bb + BLK - 1) // BLK)[: B * bb]
                          for c in range(C)]).reshape(C, B, bb)
    assert raw.size == n
    for run_len in (0, 2, 3):
        rx = api.Rx(C)
        rx.set_mode(api.WBFM)
        rx.debug_set_stream(stream)
        rx.debug_set_run_len(run_len)
        got = rx.process_block(raw, B, want_iq256=dump)
        for c in range(C):
            o = oracle.rx()
            o.set_mode(WBFM)
            for b in range(B):
                p, m, _, i = o.process(raw[c, b])
                assert (got[0][c, b, :len(p)] == p).all(), (stream, bb, run_len, c, b)
                assert int(got[2][c, b]) == m
                if dump:
                    assert (got[4][c, b] == i).all(), (stream, bb, run_len, c, b)
        assert rx.debug_counters()[5] == 0, "a launch was not committed (replayed instead)"


@pytest.mark.parametrize("C,B,bb", [(5, 17, 65536), (1, 2, 262144), (9, 33, 32768)])
def test_wbfm_batches_with_ragged_runs(oracle, C, B, bb):
    """block counts that leave a short last run (17 = 16 + 1, 33 = 2 x 16 + 1: a run of ONE block that
    re-derives its history), channel counts that are not a multiple of 8 (idle workgroups), two calls"""
    raw = np.concatenate([synth.make_input("lcg" if c % 2 else "fmtone", 200 + c, (2 * B * bb + BLK - 1) // BLK)[: 2 * B * bb]
                          for c in range(C)]).reshape(C, 2 * B, bb)
    rx = api.Rx(C)
    rx.set_mode(api.WBFM)
    got = [rx.process_block(np.ascontiguousarray(raw[:, :B]), B), rx.process_block(np.ascontiguousarray(raw[:, B:]), B)]
    pcm = np.concatenate([g[0] for g in got], axis=1)
    mag = np.concatenate([g[2] for g in got], axis=1)
    for c in range(C):
        o = oracle.rx()
        o.set_mode(WBFM)
        for b in range(2 * B):
            p, m, _, _ = o.process(raw[c, b])
            assert (pcm[c, b, :len(p)] == p).all(), (C, B, bb, c, b)
            assert int(mag[c, b]) == m
    assert rx.debug_counters()[5] == 0


@pytest.mark.parametrize("C,B,bb", [(3, 61, 32768), (2, 64, 32768), (2, 40, 65536), (2, 100, 32768)])
def test_wbfm_long_runs_of_short_blocks(oracle, C, B, bb):
    """ONE run per channel (the hook: what a bank of 256 channels or more gets) of far more than 16 blocks of the
    shortest sizes the flow kernel takes (four and eight units): the per-block squelch sums live in sixteen slots, and
    sixteen blocks of four units are no more than the ring spans -- a unit of block b waits for block b - 16 to be
    finished (round 5; the stress build finds the hole without the wait within seconds).  Two calls."""
    raw = np.concatenate([synth.make_input("lcg" if c % 2 else "fmtone", 300 + c, (2 * B * bb + BLK - 1) // BLK)[: 2 * B * bb]
                          for c in range(C)]).reshape(C, 2 * B, bb)
    rx = api.Rx(C)
    rx.set_mode(api.WBFM)
    rx.debug_set_run_len(64)
    got = [rx.process_block(np.ascontiguousarray(raw[:, :B]), B), rx.process_block(np.ascontiguousarray(raw[:, B:]), B)]
    pcm = np.concatenate([g[0] for g in got], axis=1)
    mag = np.concatenate([g[2] for g in got], axis=1)
    for c in range(C):
        o = oracle.rx()
        o.set_mode(WBFM)
        for b in range(2 * B):
            p, m, _, _ = o.process(raw[c, b])
            assert (pcm[c, b, :len(p)] == p).all(), (C, B, bb, c, b)
            assert int(mag[c, b]) == m
    assert rx.debug_counters()[5] == 0, "a launch was not committed"


def test_setters_from_a_second_thread(oracle):
    """SURVEY 8b, threading: acceptIqData has one caller thread, the setters arrive unsynchronised from the CLI
    thread (reference: the telnet thread calls Radio's setters while DataConsumer's thread demodulates).  A second
    thread hammers set_gain / set_threshold / set_mode / reset-free setters with the values already in force (so
    that the expected PCM is known) while the first one processes blocks, single and batched: no crash, no torn
    configuration -- every block equals the oracle's -- and a real gain change issued from the second thread
    between two calls takes effect at the next block."""
    import threading
    C, B = 3, 10
    xs = np.stack([synth.make_input("fmtone" if c else "lcg", 90 + c, B + 4) for c in range(C)]).reshape(C, B + 4, BLK)
    g0 = float(np.float32(256000 / (2 * np.pi)))
    rx = api.Rx(C)
    rx.set_mode(api.WBFM)
    stop = threading.Event()
    calls = [0]

    def hammer():
        k = 0
        while not stop.is_set():
            rx.set_gain(api.WBFM, g0, channel=k % C)
            rx.set_threshold(-200)
            rx.set_mode(api.WBFM, channel=(k + 1) % C)
            k += 1
        calls[0] = k

    t = threading.Thread(target=hammer)
    t.start()
    got = [rx.process_block(xs[:, b:b + 1], 1)[0] for b in range(B - 4)]
    got.append(rx.process_block(xs[:, B - 4:B], 4)[0])
    stop.set()
    t.join()
    assert calls[0] > 10
    pcm = np.concatenate(got, axis=1)
    # a real change from another thread, between two calls
    t2 = threading.Thread(target=lambda: rx.set_gain(api.WBFM, g0 * 0.5))
    t2.start()
    t2.join()
    tail = rx.process_block(xs[:, B:B + 4], 4)[0]
    for c in range(C):
        o = oracle.rx()
        o.set_mode(WBFM)
        for b in range(B):
            p = o.process(xs[c, b])[0]
            assert (pcm[c, b, :len(p)] == p).all(), (c, b)
        o.set_gain(WBFM, g0 * 0.5)
        for b in range(4):
            p = o.process(xs[c, B + b])[0]
            assert (tail[c, b, :len(p)] == p).all(), (c, B + b)


import os as _os


@pytest.mark.parametrize("seed", list(range(1, 1 + int(_os.environ.get("HRFD_WALK_SEEDS", "8")))))
def test_random_walk_of_calls_against_the_oracle(oracle, seed):
    """A differential walk nobody chose: a bank of 3..70 channels in random modes (NONE included), and a sequence of
    calls with random block counts (1 = the reference's cadence and the exact per-block kernels, 2..20 = the batch
    kernels), block sizes (whole units, half units, the 1024-byte minimum), input kinds, and -- between calls -- mode
    switches, gain changes and squelch thresholds that do and do not close gates.  Every channel against its own
    sequential oracle: PCM, n_pcm, magnitude, signal_allowed of every block, bit for bit, state carried across
    everything."""
    rng = np.random.default_rng(1000 + seed)
    C = int(rng.integers(3, 71))
    all_modes = [NONE, AM, FM, WBFM, LSB, USB]
    modes = [all_modes[int(rng.integers(0, 6))] for _ in range(C)]
    if seed % 2:
        modes = [WBFM if m in (AM, NONE) else m for m in modes]     # (odd seeds: mostly the flow kernels' banks)
    rx = api.Rx(C)
    orc = []
    for c in range(C):
        rx.set_mode(modes[c], channel=c)
        o = oracle.rx(); o.set_mode(modes[c]); orc.append(o)
    kinds = ["fmtone", "lcg", "amtone", "zeros", "dc_neg", "dc_pos", "impulse"]
    seeds = [int(rng.integers(0, 10000)) for _ in range(C)]
    ckind = [kinds[int(rng.integers(0, len(kinds)))] if rng.random() < 0.5 else "fmtone" for _ in range(C)]
    for call in range(5):
        bb = int(rng.choice([262144, 262144, 131072, 65536, 36864, 258048, 8192, 1024]))
        B = int(rng.choice([1, 1, 2, 3, 5, 16, 20])) if bb >= 8192 else int(rng.choice([1, 2, 7]))
        need = (B * bb + BLK - 1) // BLK
        xs = np.stack([synth.make_input(ckind[c], seeds[c] + 17 * call, need)[:B * bb].reshape(B, bb) for c in range(C)])
        # between calls: what the CLI thread does to a running radio (diagUi.cc: set demodmode / gains / squelch)
        for c in range(C):
            r = rng.random()
            if r < 0.10:
                modes[c] = all_modes[int(rng.integers(1, 6))]
                rx.set_mode(modes[c], channel=c); orc[c].set_mode(modes[c])
            elif r < 0.20 and modes[c] != NONE:
                g = float(rng.choice([1.0, 300.0, 4000.0, 40743.6, 2.5e5]))
                gm = LSB if modes[c] == USB else modes[c]
                rx.set_gain(gm, g, channel=c); orc[c].set_gain(gm, g)
            elif r < 0.30:
                t = int(rng.choice([-200, -60, -30, -22, -10]))
                rx.set_threshold(t, channel=c); orc[c].set_threshold(t)
        pcm, n_pcm, mag, allowed, _ = rx.process_block(xs, B)
        for c in range(C):
            for b in range(B):
                p, m, a, _ = orc[c].process(xs[c, b])
                assert n_pcm[c, b] == len(p) and int(mag[c, b]) == m and bool(allowed[c, b]) == a, (seed, call, c, b, modes[c], bb, B)
                assert (pcm[c, b, :len(p)] == p).all(), (seed, call, c, b, modes[c], bb, B)


@pytest.mark.parametrize("seed", list(range(1, 1 + int(_os.environ.get("HRFD_WALK_SEEDS", "8")))))
def test_random_walk_of_long_batches(oracle, seed):
    """The same kind of walk where the first one does not go: batches of 17..80 blocks (more than 64: several runs per WBFM channel, the FIR modes on their block kernels), the shortest blocks the flow kernels
    take (32 KiB = four units) among them, small banks FORCED onto the shapes a large bank gets (one run of up to 64 blocks
    per channel, the FIR modes on the flow kernel, several kinds in ONE launch of k_rx_flow_bank), three calls, state carried
    across them; one walk in four with squelch thresholds that close gates inside the batches, the iq dump on in a third of the calls.  A walk without thresholds
    and with one run per channel must not have needed a single replay (the stress build of the code before the block-slot
    wait fails 4 of the first 24 walks on exactly that: profiles/r5_walk_long.txt)."""
    rng = np.random.default_rng(5000 + seed)
    C = int(rng.integers(2, 11))
    all_modes = [NONE, AM, FM, WBFM, LSB, USB]
    modes = [all_modes[int(rng.integers(1, 6))] if rng.random() < 0.9 else NONE for _ in range(C)]
    if seed % 3 == 0:
        modes = [WBFM] * C
    rx = api.Rx(C)
    run_len = int(rng.choice([0, 16, 64, 64]))
    fir_shape = int(rng.choice([1, 1, 2, -1]))
    from tests.hooks import HOOKS_ON
    if HOOKS_ON:                                           # (the shipped state walks the same batches on the dispatch a user gets)
        rx.debug_set_run_len(run_len)
        rx.debug_set_fir_flow(fir_shape)
    else:
        run_len = 0                                        # (automatic: a small bank is cut into several runs per channel)
    gates = seed % 4 == 0
    orc = []
    for c in range(C):
        rx.set_mode(modes[c], channel=c)
        o = oracle.rx(); o.set_mode(modes[c]); orc.append(o)
    kinds = ["fmtone", "lcg", "amtone", "zeros", "dc_neg", "impulse"]
    seeds = [int(rng.integers(0, 10000)) for _ in range(C)]
    ckind = [kinds[int(rng.integers(0, len(kinds)))] if rng.random() < 0.5 else "fmtone" for _ in range(C)]
    cut = False                                            # a WBFM channel cut into several runs in some call
    for call in range(3):
        bb = int(rng.choice([32768, 32768, 65536, 131072, 262144]))
        B = int(rng.choice([17, 24, 33, 47, 64, 80]))
        B = min(B, (8 << 20) // bb)                        # (at most 8 MiB per channel and call: the oracle's time)
        cut = cut or B > 64
        need = (B * bb + BLK - 1) // BLK
        xs = np.stack([synth.make_input(ckind[c], seeds[c] + 17 * call, need)[:B * bb].reshape(B, bb) for c in range(C)])
        for c in range(C):
            r = rng.random()
            if r < 0.15 and modes[c] != NONE:
                g = float(rng.choice([1.0, 300.0, 4000.0, 40743.6]))
                gm = LSB if modes[c] == USB else modes[c]
                rx.set_gain(gm, g, channel=c); orc[c].set_gain(gm, g)
            elif r < 0.45 and gates:
                t = int(rng.choice([-200, -60, -30, -22]))
                rx.set_threshold(t, channel=c); orc[c].set_threshold(t)
        dump = rng.random() < 0.3                          # (`enable iqdump`: the DUMP builds of the kernels, the 256 kS/s stream out as well)
        pcm, n_pcm, mag, allowed, iq256 = rx.process_block(xs, B, want_iq256=dump)
        for c in range(C):
            for b in range(B):
                p, m, a, d = orc[c].process(xs[c, b])
                assert n_pcm[c, b] == len(p) and int(mag[c, b]) == m and bool(allowed[c, b]) == a, (seed, call, c, b, modes[c], bb, B)
                assert (pcm[c, b, :len(p)] == p).all(), (seed, call, c, b, modes[c], bb, B)
                if dump:
                    assert (iq256[c, b] == d).all(), (seed, call, c, b, modes[c], bb, B)
    if not gates and run_len == 64 and not cut:
        # (a WBFM channel cut into several runs speculates across the cuts: on inputs like a full-scale DC the history
        #  re-derived in front of a run is not the stream's, the check says so and the channel is replayed -- by design)
        assert rx.debug_counters()[5] == 0, ("a launch was not committed", seed)
