"""Short and odd-sized blocks on the GPU (k_rx_ragged, hrfd_rx_ragged.hip) through the C ABI: against the fixtures the
compiled reference produced (tests/golden/make_golden_short.py) and against the CPU oracle on seeded inputs.

The reference takes any byteCount: DataConsumer::acceptData passes short USB transfers on (DataConsumer.cc:229-241,
:341-343), every decimator keeps its commutator position between calls (Decimator_int16.cc:321-362)."""
import os
import subprocess

import numpy as np
import pytest

from hackrfdiags_amd import api, synth
from tests import shortcheck as S
from tests.reflib import AM, FM, WBFM, LSB, USB, NONE

pytestmark = pytest.mark.gpu
BLK = synth.BLOCK_BYTES
ARR, MAN = S.load()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def engine():
    if api.device_count() < 1:
        pytest.fail("no GPU visible: the HIP path cannot run (and there is no CPU fallback)")
    return api.Engine()


# ---------------------------------------------------------------- the reference's own outputs
@pytest.mark.parametrize("case", MAN["rx"], ids=lambda c: c["key"])
def test_golden_short_block_sequences(engine, case):
    """(261632, 262144), (16896, 245248), (512, 261632), (1000, 262144), a walk off and on the 512-byte grid, calls of a
    few bytes: per call the PCM count and samples, signalMagnitude and the iq dump of the compiled reference"""
    S.check_rx_sequence(engine, ARR, case)


@pytest.mark.parametrize("case", MAN["squelch"], ids=lambda c: c["key"])
def test_golden_squelch_over_short_blocks(engine, case):
    S.check_squelch(engine, ARR, case)


@pytest.mark.parametrize("case", MAN["demod"], ids=lambda c: c["key"])
def test_golden_inner_api_with_uneven_byte_counts(engine, case):
    S.check_demod(engine, ARR, case)


# ---------------------------------------------------------------- against the oracle
def _walk(oracle, rx, xs, sizes, modes, thresholds=None, n_blocks=None, events=None):
    """one bank, call by call: sizes[i] bytes per block and channel (n_blocks[i] blocks in one call), every channel its
    own input and mode; events: {call: fn(rx, orcs)} applied in front of a call"""
    C = xs.shape[0]
    orcs = []
    for c in range(C):
        o = oracle.rx()
        o.set_mode(modes[c])
        if thresholds is not None:
            o.set_threshold(thresholds[c])
        orcs.append(o)
    off = 0
    for i, n in enumerate(sizes):
        nb = 1 if n_blocks is None else n_blocks[i]
        if events and i in events:
            events[i](rx, orcs)
        pending = rx.pending_samples()
        pcm, n_pcm, mag, allowed, dump = rx.process_block(xs[:, off:off + n * nb].reshape(C, nb, n), nb, want_iq256=True)
        for c in range(C):
            held = pending
            for b in range(nb):
                wp, wm, wa, wd = orcs[c].process(xs[c, off + b * n:off + (b + 1) * n])
                assert n_pcm[c, b] == len(wp), (i, n, c, b, int(n_pcm[c, b]), len(wp))
                assert (pcm[c, b, :len(wp)] == wp).all(), (i, n, c, b)
                assert mag[c, b] == wm and bool(allowed[c, b]) == wa, (i, n, c, b)
                cnt = 2 * ((held + n // 2) // 8)
                held = (held + n // 2) % 8
                assert cnt == len(wd) and (dump[c, b, :cnt] == wd).all(), (i, n, c, b)
        off += n * nb
    return orcs


@pytest.mark.parametrize("mode", [AM, FM, WBFM, LSB, USB])
def test_blocks_some_usb_packets_short_pass_through_and_the_stream_is_back_on_the_streaming_kernels(oracle, mode):
    """multiples of 512 that are not multiples of 1024 stay ON the grid: k_rx_ragged reads and writes ChanState in the
    streaming kernels' format (pipelines re-created from the demodulator's input tail), so full blocks before, between
    and after run on the streaming kernels -- batches included -- and everything equals the sequential oracle"""
    C = 5
    sizes = [262144, 261632, 262144, 512, 262144, 1536, 258560, 262144, 262144]
    nbl = [2, 1, 1, 1, 3, 2, 1, 2, 1]
    total = sum(s * b for s, b in zip(sizes, nbl))
    xs = np.stack([synth.make_input("amtone" if c % 2 else "fmtone", 400 + 10 * mode + c, (total + BLK - 1) // BLK)[:total]
                   for c in range(C)])
    rx = api.Rx(C)
    rx.set_mode(mode)
    _walk(oracle, rx, xs, sizes, [mode] * C, n_blocks=nbl)
    assert rx.pending_samples() == 0
    assert rx.debug_ragged() == (False, 4)                  # on the grid; the four short calls ran on k_rx_ragged


def test_bank_of_every_mode_off_the_grid(oracle):
    """a bank with every mode (and a channel without a demodulator), lengths that leave the 512-byte grid at once, calls
    of several blocks, gain changes, a mode switch and a squelch threshold on the way"""
    modes = [WBFM, AM, FM, LSB, USB, NONE, WBFM, FM, AM]
    C = len(modes)
    sizes = [1000, 262144, 30, 4098, 262144, 512, 18, 131070, 16, 262142, 261632, 262144, 2050]
    nbl = [1, 2, 1, 3, 1, 1, 1, 2, 1, 1, 1, 2, 5]
    total = sum(s * b for s, b in zip(sizes, nbl))
    xs = np.stack([synth.make_input("fmtone" if c % 3 else "amtone", 500 + c, (total + BLK - 1) // BLK)[:total]
                   for c in range(C)])
    rx = api.Rx(C)
    for c, m in enumerate(modes):
        rx.set_mode(m, c)

    def gains(rx, orcs):
        for c, m in enumerate(modes):
            if m != NONE:
                rx.set_gain(m, 777.0 + c, c)
                orcs[c].set_gain(m, 777.0 + c)

    def switch(rx, orcs):
        rx.set_mode(WBFM, 1)
        orcs[1].set_mode(WBFM)
        rx.set_mode(LSB, 4)
        orcs[4].set_mode(LSB)

    def back(rx, orcs):
        rx.set_mode(AM, 1)
        orcs[1].set_mode(AM)

    _walk(oracle, rx, xs, sizes, modes, n_blocks=nbl, events={3: gains, 6: switch, 10: back})
    assert rx.debug_ragged()[0] is True


@pytest.mark.parametrize("mode", [WBFM, AM, FM, USB])
def test_gates_over_short_blocks(oracle, mode):
    """Squelch over calls of uneven length, several blocks per call: loud / quiet channels under a -30 dBFS threshold --
    the demodulator sees only the allowed calls and its commutators stand still in between"""
    sizes = [262144, 1000, 261144, 4098, 258046, 512, 261632, 262144]
    nbl = [1, 1, 1, 2, 1, 3, 1, 1]
    patterns = [[1, 1, 0, 0, 1, 0, 0, 1], [0, 0, 1, 1, 0, 1, 1, 0], [1, 0, 1, 0, 1, 0, 1, 0], [1, 1, 1, 1, 1, 1, 1, 1]]
    C = len(patterns)
    total = sum(s * b for s, b in zip(sizes, nbl))
    rows = []
    for c in range(C):
        loud = synth.make_input("fmtone", 600 + c, (total + BLK - 1) // BLK)[:total]
        off = 0
        row = np.zeros(total, np.int8)
        for i, (s, b) in enumerate(zip(sizes, nbl)):
            if patterns[c][i]:
                row[off:off + s * b] = loud[off:off + s * b]
            off += s * b
        rows.append(row)
    xs = np.stack(rows)
    rx = api.Rx(C)
    rx.set_mode(mode)
    rx.set_threshold(-30)
    _walk(oracle, rx, xs, sizes, [mode] * C, thresholds=[-30] * C, n_blocks=nbl)


@pytest.mark.parametrize("mode", [AM, FM, WBFM, LSB])
def test_inner_api_off_the_grid_with_reset_and_gain(oracle, mode):
    """X::acceptIqData with uneven byte counts, X::resetDemodulator in between (Decimator_int16::resetFilterState clears
    the pipeline AND the commutator position) and a gain change"""
    sizes = [32768, 62, 32768, 2, 130, 64, 32766, 1000, 32768, 4, 32768]
    x = synth.make_input("lcg", 31 + mode, 2)[:sum(sizes)]
    d, o = api.Demod(mode, 1), oracle.demod(mode)
    off = 0
    for i, n in enumerate(sizes):
        if i == 4:
            d.set_gain(555.0)
            o.set_gain(555.0)
        if i in (6, 9):
            d.reset()
            o.reset()
        got, want = d.process(x[off:off + n]), o.process(x[off:off + n])
        assert len(got) == len(want) and (got == want).all(), (i, n)
        off += n


def test_ingest_with_a_short_block_length(oracle):
    """the block transport with blocks one USB packet short (hrfd_ingest_* over hrfd_rx_process_device)"""
    C, B, bb = 6, 2, 261632
    rx = api.Rx(C)
    rx.set_mode(WBFM)
    ing = api.Ingest(rx, bb, B, 2)
    xs = np.stack([synth.make_input("fmtone", 700 + c, 4)[:4 * bb] for c in range(C)]).reshape(C, 4, bb)
    outs = []
    for k in range(2):
        ing.acquire()[...] = xs[:, k * B:(k + 1) * B]
        ing.submit(0)
    for k in range(2):
        outs.append(ing.collect())
    for c in range(C):
        o = oracle.rx()
        o.set_mode(WBFM)
        for k in range(2):
            for b in range(B):
                wp, wm, wa, _ = o.process(xs[c, k * B + b])
                assert outs[k][1][c, b] == len(wp) and (outs[k][0][c, b, :len(wp)] == wp).all(), (c, k, b)
                assert outs[k][2][c, b] == wm


def test_sizes_the_reference_cannot_take_either_are_refused():
    """odd counts (the reference's Q loop reads bufferPtr[byteCount], IqDataProcessor.cc:474), nothing at all, and more
    than its arrays hold (DataConsumer clips to 262144 before the call, DataConsumer.cc:229-233)"""
    rx = api.Rx(1)
    for n in (1001, 2 * 262144):
        with pytest.raises(api.HrfdError):
            rx.process_block(np.zeros((1, 1, n), dtype=np.int8), 1)
    with pytest.raises(api.HrfdError):
        api.Demod(WBFM, 1).process(np.zeros(32770, dtype=np.int8))
    # a call that completes no 256 kS/s sample: the reference divides by zero (SignalDetector.cc:255); here magnitude 0
    pcm, n_pcm, mag, allowed, _ = rx.process_block(np.full((1, 1, 6), 100, dtype=np.int8), 1)
    assert n_pcm[0, 0] == 0 and mag[0, 0] == 0
    assert rx.pending_samples() == 3


# ---------------------------------------------------------------- the shim classes
def _demo():
    from tests.test_shim import _build_demo, DEMO
    _build_demo()
    return DEMO


@pytest.mark.parametrize("mode", [AM, FM, WBFM, LSB, USB])
def test_shim_takes_the_byte_counts_the_reference_takes(oracle, mode):
    """IqDataProcessor::acceptIqData of the shim with short, uneven, odd and oversized counts: never an abort, the PCM
    callbacks of the reference call by call (a callback with a count of 0 included)"""
    demo = _demo()
    sizes = [261632, 262144, 1000, 262144, 30, 4098, 1001, 262144]
    x = synth.make_input("fmtone", 9, 6)
    env = dict(os.environ, HRFD_DEMO_COUNTS="1")
    r = subprocess.run([demo, str(mode), "outer", ",".join(str(s) for s in sizes)], input=x[:sum(sizes)].tobytes(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
    got = np.frombuffer(r.stdout, dtype=np.int16)
    counts = [int(l.split()[1]) for l in r.stderr.decode().splitlines() if l.startswith("pcm ")]
    o = oracle.rx()
    o.set_mode(mode)
    want, wcounts, off = [], [], 0
    for n in sizes:
        # the odd count: one byte further, like the reference's Q loop (IqDataProcessor.cc:474); the demo leaves a 0 there
        p = o.process(np.concatenate([x[off:off + n], np.zeros(n & 1, np.int8)]))[0]
        off += n
        want.append(p)
        wcounts.append(len(p))
    assert counts == wcounts
    assert (got == np.concatenate(want)).all()
