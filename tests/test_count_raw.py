"""The reference's OWN test input, signals/count.raw (five seconds of 8 kS/s PCM: what signals/makeThem.sh feeds through
`./a.out < count.raw | ./interpolateSignal`, and what README.txt:117-136 runs through the modulator harnesses to make test
vectors for the demodulators), through everything on the transmit side and back through the receive side:

  makethem   signals/{am,dsb,pm,fm}.cc | interpolateSignal          -> int8 IQ at 2.048 MS/s
  modulator  {Ssb,Am,Fm,WbFm}Modulator::acceptData, 512 per call    -> int8 IQ   (= the reference's own harness programs
             SsbModulator/ssb.cc, AmModulator/am.cc, FmModulator/fm.cc, WbFmModulator/wbfm.cc < count.raw)
  loop       that IQ, 64 kHz down (the radio's tuning offset, Radio.cc:1187-1191), through IqDataProcessor::acceptIqData in
             the matching demodulator mode -> PCM: the closed loop of README.txt:133-136

Expected values: tests/golden/golden_count.* = the reference's own sources compiled in place (tests/golden/
make_golden_count.py).  CPU: the oracle reproduces every digest.  GPU: libhrfd reproduces them -- bit for bit, every one:
where the reference goes through libm cosf / sinf (FM modulator, pm / fm generators) the device runs glibc's algorithm
restated (round 5; rounds 1-4: +-1 LSB there)."""
import json
import os

import numpy as np
import pytest

from hackrfdiags_amd import synth
from tests import toolsupport as T
from tests.reflib import AM, FM, WBFM, LSB

HERE = os.path.dirname(os.path.abspath(__file__))
ARR = np.load(os.path.join(HERE, "golden", "golden_count.npz"))
MAN = json.load(open(os.path.join(HERE, "golden", "golden_count.json")))
PCM = np.fromfile(os.path.join(HERE, "golden", "count.raw"), dtype="<i2")
BLK = synth.BLOCK_BYTES
MODE = {"ssb": LSB, "am": AM, "fm": FM, "wbfm": WBFM}


def test_the_input_is_the_references_file():
    assert PCM.size == MAN["input"]["samples"] == 40000 and synth.digest(PCM) == MAN["input"]["sha256"]


def _oracle_mod(oracle, kind):
    return oracle.ssbmod(True) if kind == "ssb" else getattr(oracle, kind + "mod")()


def _oracle_modulate(oracle, kind):
    m = _oracle_mod(oracle, kind)
    return np.concatenate([m.process(PCM[s:s + 512]) for s in range(0, PCM.size, 512)])


def _lsb_diff(a, b):
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    return np.minimum(d, 256 - d)


# ------------------------------------------------------------------------------------------------ CPU: the oracle
@pytest.mark.parametrize("case", MAN["makethem"], ids=lambda c: c["kind"])
def test_oracle_makethem(oracle, case):
    pairs, _ = T.orc_siggen(oracle, case["kind"], PCM)
    iq = oracle.interp().process(pairs)
    assert iq.size == case["iq_bytes"] and synth.digest(iq) == case["iq_sha256"]
    assert (iq[:2048] == ARR[f"makethem_{case['kind']}_head"]).all() and (iq[-2048:] == ARR[f"makethem_{case['kind']}_tail"]).all()


@pytest.mark.parametrize("case", MAN["modulator"], ids=lambda c: c["kind"])
def test_oracle_modulators(oracle, case):
    iq = _oracle_modulate(oracle, case["kind"])
    assert iq.size == case["iq_bytes"] and synth.digest(iq) == case["iq_sha256"]
    # ... which is also what the reference's own harness program of this modulator writes for count.raw on its stdin
    assert case["program"].endswith(".cc") and case["program_sha256"] == case["iq_sha256"]


@pytest.mark.parametrize("case", MAN["loop"], ids=lambda c: c["kind"])
def test_oracle_closed_loop(oracle, case):
    air = T.retune_minus_64k(_oracle_modulate(oracle, case["kind"]))
    assert synth.digest(air) == case["air_sha256"], "the channel model (numpy) gave other bytes on this platform"
    rx = oracle.rx()
    rx.set_mode(case["mode"])
    back = np.concatenate([rx.process(air[s:s + BLK])[0] for s in range(0, air.size, BLK)])
    assert back.size == case["pcm_samples"] and (back == ARR[f"loop_{case['kind']}_pcm"]).all()
    assert synth.digest(back) == case["pcm_sha256"]
    # and it IS the author's audio that comes back (FM 0.99, WBFM 0.93, AM 0.69, SSB 0.60 in the reference's own chain)
    assert case["best_abs_correlation_with_count_raw"] > 0.55


# ------------------------------------------------------------------------------------------------ GPU: libhrfd
@pytest.mark.gpu
@pytest.mark.parametrize("case", MAN["makethem"], ids=lambda c: c["kind"])
def test_gpu_makethem(oracle, case):
    from hackrfdiags_amd import api
    kind = case["kind"]
    m = api.Mod({"am": api.MOD_SIG_AM, "dsb": api.MOD_SIG_DSB, "pm": api.MOD_SIG_PM, "fm": api.MOD_SIG_FM}[kind], 1)
    iq = np.atleast_2d(m.process(PCM.reshape(1, -1)))[0]
    assert iq.size == case["iq_bytes"]
    if kind in ("am", "dsb"):
        assert synth.digest(iq) == case["iq_sha256"]
    else:                                                   # cos / sin: libm cosf / sinf in the reference
        pairs, _ = T.orc_siggen(oracle, kind, PCM)
        assert synth.digest(iq) == case["iq_sha256"]       # (round 5: glibc's cosf / sinf restated on the device -- the reference's bytes)
        assert (iq == oracle.interp().process(pairs)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("case", MAN["modulator"], ids=lambda c: c["kind"])
def test_gpu_modulators(oracle, case):
    from hackrfdiags_amd import api
    kind = case["kind"]
    m = api.Mod({"ssb": api.MOD_SSB, "am": api.MOD_AM, "fm": api.MOD_FM, "wbfm": api.MOD_WBFM}[kind], 1)
    iq = np.atleast_2d(m.process(PCM.reshape(1, -1)))[0]    # ONE call of 40000 samples (the reference: 79 of <= 512)
    assert iq.size == case["iq_bytes"]
    assert synth.digest(iq) == case["iq_sha256"]           # every kind, FM included since round 5 (glibc's cosf / sinf on the device)


def _gpu_demodulate(api, mode, air):
    """78 whole blocks as two batches of 39 (the flow kernel), then the 32768-byte rest (the block kernel)"""
    rx = api.Rx(1)
    rx.set_mode(mode)
    whole = air.size // BLK
    out = []
    for lo in range(0, whole, 39):
        n = min(39, whole - lo)
        out.append(rx.process_block(air[lo * BLK:(lo + n) * BLK].reshape(1, n, BLK), n)[0].reshape(-1))
    rest = air[whole * BLK:]
    if rest.size:
        out.append(rx.process_block(rest.reshape(1, 1, -1), 1)[0].reshape(-1))
    return np.concatenate(out)


@pytest.mark.gpu
@pytest.mark.parametrize("case", MAN["loop"], ids=lambda c: c["kind"])
def test_gpu_closed_loop(oracle, case):
    """GPU modulator -> the 64 kHz channel -> GPU demodulator = the reference's loop, bit for bit, in all four kinds (FM
    since round 5: the GPU's FM signal IS the reference's)."""
    from hackrfdiags_amd import api
    kind = case["kind"]
    m = api.Mod({"ssb": api.MOD_SSB, "am": api.MOD_AM, "fm": api.MOD_FM, "wbfm": api.MOD_WBFM}[kind], 1)
    mine = np.atleast_2d(m.process(PCM.reshape(1, -1)))[0]
    air = T.retune_minus_64k(mine)
    assert synth.digest(air) == case["air_sha256"]
    back = _gpu_demodulate(api, case["mode"], air)
    assert back.size == case["pcm_samples"] and (back == ARR[f"loop_{kind}_pcm"]).all()
