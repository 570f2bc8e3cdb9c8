"""Checks of the short / odd-sized block fixtures (tests/golden/golden_short.*, made by the compiled reference:
tests/golden/make_golden_short.py), parameterised over an engine like tests/goldencheck.py: the CPU oracle
(-m "not gpu") and the HIP path (-m gpu) run the same assertions.

engine.rx() -> object with set_mode / set_threshold / process(iq) -> (pcm, magnitude, allowed, iq256 of the call's own count)
engine.demod(mode) -> object with process(iq256) -> pcm"""
import json
import os
import sys

import numpy as np

from hackrfdiags_amd import synth

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, HERE)
import make_golden_short as gen  # noqa: E402  (the sequences and their inputs; main() needs the reference, the rest does not)


def load():
    arrays = np.load(os.path.join(HERE, "golden_short.npz"))
    with open(os.path.join(HERE, "golden_short.json")) as f:
        manifest = json.load(f)
    return arrays, manifest


def check_rx_sequence(engine, arrays, case):
    x = gen.sequence_input(case["sequence"], case["kind"], case["seed"])
    assert synth.digest(x) == case["input_sha256"], "input generator drifted"
    h = engine.rx()
    h.set_mode(case["mode"])
    want = arrays[case["key"] + "_pcm"]
    o = w = 0
    for i, n in enumerate(case["sizes"]):
        pcm, mag, _, dump = h.process(x[o:o + n])
        o += n
        cnt = case["counts"][i]
        assert len(pcm) == cnt, (case["key"], i, n, len(pcm), cnt)
        assert (pcm == want[w:w + cnt]).all(), (case["key"], i, n)
        w += cnt
        assert mag == case["mags"][i], (case["key"], i, n, mag)
        assert len(dump) == case["dump_bytes"][i], (case["key"], i, n, len(dump))
        assert synth.digest(np.ascontiguousarray(dump)) == case["dump_sha256"][i], (case["key"], i, n)
        if i == 0 and case["mode"] == 3:
            assert (dump == arrays[f"short_{case['sequence']}_dump0"]).all()


def squelch_input(case):
    sizes, pattern = case["sizes"], case["pattern"]
    loud = synth.make_input("fmtone", 3, len(sizes))
    return np.concatenate([loud[sum(sizes[:i]):sum(sizes[:i + 1])] if bit else np.zeros(sizes[i], np.int8)
                           for i, bit in enumerate(pattern)])


def check_squelch(engine, arrays, case):
    x = squelch_input(case)
    h = engine.rx()
    h.set_mode(case["mode"])
    h.set_threshold(case["threshold"])
    want = arrays[case["key"] + "_pcm"]
    o = w = 0
    for i, n in enumerate(case["sizes"]):
        pcm, mag, _, _ = h.process(x[o:o + n])
        o += n
        cnt = case["counts"][i]
        assert len(pcm) == cnt and (pcm == want[w:w + cnt]).all(), (case["key"], i, n, len(pcm), cnt)
        w += cnt
        assert mag == case["mags"][i], (case["key"], i)


def check_demod(engine, arrays, case):
    sizes = case["sizes"]
    x = synth.make_input(case["kind"], case["seed"], 1)[:sum(sizes)]
    d = engine.demod(case["mode"])
    want = arrays[case["key"] + "_pcm"]
    o = w = 0
    for i, n in enumerate(sizes):
        pcm = d.process(x[o:o + n])
        o += n
        cnt = case["counts"][i]
        assert len(pcm) == cnt and (pcm == want[w:w + cnt]).all(), (case["key"], i, n, len(pcm), cnt)
        w += cnt
