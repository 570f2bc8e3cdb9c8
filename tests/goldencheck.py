"""Golden-vector checks, parameterised over an 'engine' so the same assertions
serve the CPU oracle (-m "not gpu") and the HIP path (-m gpu).

An engine provides:  rx() -> object with set_mode/set_gain/set_threshold/process(iq)->(pcm,mag,allowed,iq256)
                     ssbmod(lsb) -> object with process(pcm)->iq ; interp() -> process(iq16)->iq
"""
import json
import os

import numpy as np

from hackrfdiags_amd import synth

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BLK = synth.BLOCK_BYTES


def load():
    arrays = np.load(os.path.join(HERE, "golden.npz"))
    with open(os.path.join(HERE, "golden.json")) as f:
        manifest = json.load(f)
    return arrays, manifest


def check_rx_case(engine, arrays, case, tol=0):
    x = synth.make_input(case["kind"], case["seed"], case["blocks"])
    if "input_sha256" in case:
        assert synth.digest(x) == case["input_sha256"], "input generator drifted"
    h = engine.rx()
    h.set_mode(case["mode"])
    if "gain" in case:
        h.set_gain(case["mode"], case["gain"])
    want = arrays[case["key"] if case.get("gain_case") else case["key"] + "_pcm"]
    for b in range(case["blocks"]):
        pcm, mag, _, _ = h.process(x[b * BLK:(b + 1) * BLK])
        w = want[b * 512:(b + 1) * 512]
        assert len(pcm) == 512
        if tol == 0:
            assert (pcm == w).all(), (case["key"], b)
        else:
            assert np.abs(pcm.astype(np.int32) - w.astype(np.int32)).max() <= tol, (case["key"], b)
        if not case.get("gain_case"):
            assert mag == int(arrays[case["key"] + "_mag"][b]), (case["key"], b)


def check_frontend_case(engine, arrays, case):
    x = synth.make_input(case["kind"], case["seed"], case["blocks"])
    assert synth.digest(x) == case["input_sha256"]
    h = engine.rx()
    got = np.concatenate([h.process(x[b * BLK:(b + 1) * BLK])[3] for b in range(case["blocks"])])
    assert (got == arrays[case["key"]]).all(), case["key"]


def check_long_case(engine, case):
    x = synth.make_input(case["kind"], case["seed"], case["blocks"])
    assert synth.digest(x) == case["input_sha256"]
    h = engine.rx()
    h.set_mode(case["mode"])
    pcm = np.concatenate([h.process(x[b * BLK:(b + 1) * BLK])[0] for b in range(case["blocks"])])
    assert synth.digest(pcm) == case["pcm_sha256"], case


def check_squelch(engine, arrays, case):
    loud = synth.make_input("fmtone", 1, 1)
    quiet = synth.zeros_iq(synth.BLOCK_IQ)
    h = engine.rx()
    h.set_mode(3)
    h.set_threshold(case["threshold"])
    lens, pcm, mags = [], [], []
    for bit in case["pattern"]:
        p, m, _, _ = h.process(loud if bit else quiet)
        lens.append(len(p)); pcm.append(p); mags.append(m)
    assert lens == case["lens"] and mags == case["mags"]
    assert (np.concatenate(pcm) == arrays[case["key"]]).all()


def check_chunked(engine, arrays, case):
    x = synth.make_input(case["kind"], case["seed"], 1)
    h = engine.rx()
    h.set_mode(case["mode"])
    c = case["chunk"]
    got = np.concatenate([h.process(x[o:o + c])[0] for o in range(0, len(x), c)])
    assert (got == arrays[case["key"]]).all(), case["key"]


def check_tx_case(engine, arrays, case):
    pcm_in = synth.lcg_pcm(case["seed"], case["blocks"] * 512)
    m = engine.ssbmod(bool(case["lsb"]))
    out = np.concatenate([m.process(pcm_in[b * 512:(b + 1) * 512]) for b in range(case["blocks"])])
    assert (out[:4096] == arrays[case["key"] + "_head"]).all()
    assert (out[-4096:] == arrays[case["key"] + "_tail"]).all()
    assert synth.digest(out) == case["iq_sha256"]


def check_interp(engine, arrays, manifest):
    a, b = manifest["interp"]
    out = engine.interp().process(arrays["interp_in"])
    assert (out[:8192] == arrays["interp_head"]).all() and synth.digest(out) == a["iq_sha256"]
    out = engine.interp().process(synth.lcg_pcm(b["lcg_seed"], 2 * b["pairs"]))
    assert synth.digest(out) == b["iq_sha256"]


def load_mod():
    arrays = np.load(os.path.join(HERE, "golden_mod.npz"))
    with open(os.path.join(HERE, "golden_mod.json")) as f:
        manifest = json.load(f)
    return arrays, manifest


def check_am_mod(engine, arrays, case):
    """AM modulator (integer cascade after a float scaling that is exact on both sides): bit-exact"""
    pcm = synth.lcg_pcm(case["seed"], case["calls"] * 512)
    m = engine.ammod()
    if case["index"] is not None:
        m.set_param(case["index"])
    out = np.concatenate([m.process(pcm[b * 512:(b + 1) * 512]) for b in range(case["calls"])])
    assert (out[:4096] == arrays[case["key"] + "_head"]).all()
    assert (out[-4096:] == arrays[case["key"] + "_tail"]).all()
    assert synth.digest(out) == case["iq_sha256"]


def check_fm_mod(engine, arrays, case, tol):
    """FM modulator: tol = 0 (the digest as well) for the CPU oracle and, since round 5, for the device"""
    pcm = synth.lcg_pcm(case["seed"], sum(case["calls"]))
    m = engine.fmmod()
    if case["deviation"] is not None:
        m.set_param(case["deviation"])
    parts, off = [], 0
    for n in case["calls"]:
        parts.append(m.process(pcm[off:off + n])); off += n
    out = np.concatenate(parts)
    d = np.abs(out.astype(np.int16) - arrays[case["key"]].astype(np.int16))
    d = np.minimum(d, 256 - d)
    assert d.max() <= tol, int(d.max())
    if tol == 0:
        assert synth.digest(out) == case["iq_sha256"]
    else:
        assert (d != 0).mean() < 0.01


def check_wbfm_mod(engine, arrays, case):
    """WBFM modulator: integer cascades around a table Nco -- bit-exact on both engines"""
    pcm = synth.lcg_pcm(case["seed"], case["calls"] * 512)
    m = engine.wbfmmod()
    if case["deviation"] is not None:
        m.set_param(case["deviation"])
    out = np.concatenate([m.process(pcm[b * 512:(b + 1) * 512]) for b in range(case["calls"])])
    assert (out[:4096] == arrays[case["key"] + "_head"]).all()
    assert (out[-4096:] == arrays[case["key"] + "_tail"]).all()
    assert synth.digest(out) == case["iq_sha256"]
