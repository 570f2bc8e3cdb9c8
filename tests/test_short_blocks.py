"""Short and odd-sized blocks on the CPU: the oracle against the fixtures the compiled reference produced
(tests/golden/make_golden_short.py) and, where oracle/_ref is present, against the reference itself call by call.

The reference takes any byteCount: DataConsumer::acceptData passes short USB transfers on (DataConsumer.cc:229-241,
:341-343) and every decimator keeps its commutator position between calls (Decimator_int16.cc:321-362)."""
import numpy as np
import pytest

from hackrfdiags_amd import api, synth
from tests import shortcheck as S
from tests.reflib import AM, FM, WBFM, LSB, USB

ARR, MAN = S.load()


@pytest.mark.parametrize("case", MAN["rx"], ids=lambda c: c["key"])
def test_oracle_reproduces_the_references_short_block_sequences(oracle, case):
    S.check_rx_sequence(oracle, ARR, case)


@pytest.mark.parametrize("case", MAN["squelch"], ids=lambda c: c["key"])
def test_oracle_squelch_over_short_blocks(oracle, case):
    S.check_squelch(oracle, ARR, case)


@pytest.mark.parametrize("case", MAN["demod"], ids=lambda c: c["key"])
def test_oracle_inner_api_with_uneven_byte_counts(oracle, case):
    S.check_demod(oracle, ARR, case)


def test_decimated_byte_count_formula():
    """reduceSampleRate's return value, call by call, as the compiled reference gave it: 2 * floor((held + n) / 8) with
    `held` the IQ samples the three half-band decimators hold back -- the formula of hrfd_rx_pending_samples /
    api.iq256_capacity and of the shim's iq dump."""
    case = MAN["reduce"][0]
    held = 0
    for n, want in zip(case["sizes"], case["returns"]):
        assert 2 * ((held + n // 2) // 8) == want, (n, held)
        assert want <= api.iq256_capacity(n)
        held = (held + n // 2) % 8


def test_output_capacities_hold_every_call_of_the_fixtures():
    for case in MAN["rx"] + MAN["squelch"]:
        for n, cnt in zip(case["sizes"], case["counts"]):
            assert cnt <= api.pcm_capacity(n), (n, cnt)
    for case in MAN["demod"]:
        for n, cnt in zip(case["sizes"], case["counts"]):
            assert cnt <= (n + 63) // 64, (n, cnt)


@pytest.mark.parametrize("mode", [AM, FM, WBFM, LSB, USB])
def test_oracle_equals_reference_on_a_random_walk_of_lengths(oracle, ref, mode):
    """call for call against the compiled reference: random even lengths (every call completes at least one 256 kS/s
    sample: the reference divides by zero otherwise), a gain change and a mode excursion on the way"""
    rng = np.random.default_rng(1000 + mode)
    x = synth.make_input("fmtone", 20 + mode, 18)
    a, b = oracle.rx(), ref.rx()
    for h in (a, b):
        h.set_mode(mode)
    o = 0
    for call in range(18):
        pick = rng.integers(0, 5)
        n = int([262144, 2 * rng.integers(8, 600), 512 * rng.integers(1, 512), 2 * rng.integers(8, 131072), 16][pick])
        if call == 7:
            for h in (a, b):
                h.set_gain(mode, 1234.5)
        if call in (11, 14):
            for h in (a, b):
                h.set_mode(WBFM if call == 11 and mode != WBFM else mode)
        pa, ma, _, da = a.process(x[o:o + n])
        pb, mb, _, db = b.process(x[o:o + n])
        assert len(pa) == len(pb) and (pa == pb).all(), (call, n)
        assert ma == mb and len(da) == len(db) and (da == db).all(), (call, n)
        o += n


def test_reference_reduce_sample_rate_count(ref):
    """the count formula against the reference's own return value, live"""
    rng = np.random.default_rng(5)
    h = ref.rx()
    held = 0
    x = synth.make_input("lcg", 3, 2)
    o = 0
    for _ in range(60):
        n = int(2 * rng.integers(1, 3000))
        r, _ = h.reduce_sample_rate(x[o:o + n])
        assert r == 2 * ((held + n // 2) // 8), (n, held, r)
        held = (held + n // 2) % 8
        o += n
