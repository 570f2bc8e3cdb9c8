"""The CPU oracle against the committed golden vectors (which were produced by
the reference's own compiled sources -- tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from hackrfdiags_amd import synth
from tests import goldencheck as G

ARR, MAN = G.load()


@pytest.mark.parametrize("case", MAN["rx"], ids=lambda c: c["key"])
def test_rx(oracle, case):
    G.check_rx_case(oracle, ARR, case)


@pytest.mark.parametrize("case", MAN["frontend"], ids=lambda c: c["key"])
def test_frontend(oracle, case):
    G.check_frontend_case(oracle, ARR, case)


@pytest.mark.parametrize("case", MAN["rx_long"], ids=lambda c: f"long_mode{c['mode']}")
def test_rx_long(oracle, case):
    G.check_long_case(oracle, case)


def test_squelch(oracle):
    G.check_squelch(oracle, ARR, MAN["squelch"][0])


@pytest.mark.parametrize("case", MAN["chunked"], ids=lambda c: c["key"])
def test_chunked(oracle, case):
    G.check_chunked(oracle, ARR, case)


@pytest.mark.parametrize("case", MAN["tx"], ids=lambda c: c["key"])
def test_tx(oracle, case):
    G.check_tx_case(oracle, ARR, case)


def test_interp(oracle):
    G.check_interp(oracle, ARR, MAN)


def test_tables(oracle):
    for name, want in MAN["tables"].items():
        assert oracle.quantise(oracle.table(name)).tolist() == want, name
    assert (oracle.dbfs_table() == ARR["dbfs_table"]).all()


def test_nco(oracle):
    bits = lambda a: np.ascontiguousarray(a).view(np.uint32)
    for case in MAN["nco"]:
        n = oracle.nco(case["fs"], case["f"])
        s, c = n.tables()
        assert synth.digest(s) == case["sin_sha256"] and synth.digest(c) == case["cos_sha256"]
        i0, q0 = n.run(1000, False)
        i1, q1 = n.run(1000, True)
        assert (bits(np.stack([i0, q0])) == bits(ARR[case["key"] + "_run"])).all()
        assert (bits(np.stack([i1, q1])) == bits(ARR[case["key"] + "_fast"])).all()
    assert s[8192] != 0.0      # accumulated-phase quirk: Sin[8192] = -3.46e-4, not 0


ARR_MOD, MAN_MOD = G.load_mod()


@pytest.mark.parametrize("case", MAN_MOD["am"], ids=lambda c: c["key"])
def test_am_modulator(oracle, case):
    G.check_am_mod(oracle, ARR_MOD, case)


@pytest.mark.parametrize("case", MAN_MOD["fm"], ids=lambda c: c["key"])
def test_fm_modulator(oracle, case):
    G.check_fm_mod(oracle, ARR_MOD, case, tol=0)


@pytest.mark.parametrize("case", MAN_MOD["wbfm"], ids=lambda c: c["key"])
def test_wbfm_modulator(oracle, case):
    G.check_wbfm_mod(oracle, ARR_MOD, case)


def test_oracle_reproduces_the_references_decimate_audio_program(oracle):
    """The reference's own test program of Decimator_int16 (Filters/Int16/decimateAudio.cc: original32000.raw through an
    80-tap prototype, M = 4), run HERE by the reference itself on the head of its own input file
    (tests/golden/make_golden_decimate_audio.py).  The oracle's D(N, M, h) -- the primitive every receive chain is made of
     -- must give the program's 80 000 output samples."""
    import json
    arr = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_decimate_audio.npz"))
    man = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_decimate_audio.json")))
    x = np.zeros(320000, dtype=np.int16)                    # the program's static buffer: what the file does not fill is zero
    x[:arr["input_head"].size] = arr["input_head"]
    y = oracle.decimate(arr["taps"], 4, x)
    assert y.size == man["output_samples"] == 80000
    assert (y[:12000] == arr["output_head"]).all()
    from hackrfdiags_amd import synth
    assert synth.digest(y) == man["output_sha256"]


def _filter_program_sections():
    import json
    man = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_filter_programs.json")))
    return [pytest.param(s, id=f"{prog}-{k}") for prog, secs in man.items() for k, s in enumerate(secs)]


@pytest.mark.parametrize("sec", _filter_program_sections())
def test_oracle_prints_what_the_references_filter_programs_print(oracle, sec):
    """The reference's own smoke programs of FirFilter and IirFilter (Filters/testFirFilter.cc:25-67, testIirFilter.cc:
    26-108: impulse and step through {1,2,3,4,1,1,1,8}, {1}/{0.5} and the dc-removal pair {1,-1}/{-0.95} of the AM and SSB
    demodulators), run HERE by the reference itself (tests/golden/make_golden_filter_programs.py).  The oracle's float
    filter -- the one under every demodulator's de-emphasis, differentiator and dc removal -- printed the same way must
    give the same text."""
    if sec["denominator"]:
        y = oracle.iir(sec["numerator"], sec["denominator"], sec["input"])
    else:
        y = oracle.fir(sec["numerator"], sec["input"])
    assert ["%f" % float(v) for v in y] == sec["printed"]
