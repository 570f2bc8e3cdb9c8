"""The reference-named C++ shim classes (hackrfdiags_amd/csrc/shim): they must compile
as plain host C++ against include/hrfd.h (CPU check), and a miniature of the
reference application built on them must reproduce the oracle's PCM (GPU check)."""
import os
import subprocess

import numpy as np
import pytest

from hackrfdiags_amd import synth
from tests.reflib import AM, FM, WBFM, LSB, USB

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "hackrfdiags_amd", "csrc", "shim")
DEMO = os.path.join(ROOT, "tests", "cpp", "shim_demo")


def _build_demo():
    lib = os.path.join(ROOT, "hackrfdiags_amd", "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-o", DEMO, os.path.join(ROOT, "tests", "cpp", "shim_demo.cc"),
           os.path.join(SHIM, "hrfd_shim.cc"),
           "-I", os.path.join(ROOT, "include"), "-I", SHIM, "-I", os.path.join(ROOT, "tests", "cpp"),   # tests/cpp: the UdpClient test double
           "-L", lib, "-lhrfd", "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)


def test_shim_compiles_and_links_as_host_cxx():
    # no HIP headers, no hipcc: the shim is the reference-side binding a maintainer adds
    obj = os.path.join(ROOT, "tests", "cpp", "hrfd_shim.o")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-c", "-o", obj, os.path.join(SHIM, "hrfd_shim.cc"),
                           "-I", os.path.join(ROOT, "include"), "-I", SHIM, "-I", os.path.join(ROOT, "tests", "cpp")])
    os.remove(obj)
    _build_demo()
    assert os.path.exists(DEMO)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [AM, FM, WBFM, LSB, USB])
def test_shim_outer_boundary_reproduces_oracle(oracle, mode):
    _build_demo()
    x = synth.make_input("fmtone", 6, 3)
    out = subprocess.run([DEMO, str(mode), "outer", "262144"], input=x.tobytes(), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, check=True).stdout
    got = np.frombuffer(out, dtype=np.int16)
    o = oracle.rx(); o.set_mode(mode)
    want = np.concatenate([o.process(x[b * 262144:(b + 1) * 262144])[0] for b in range(3)])
    assert len(got) == len(want) and (got == want).all()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [AM, WBFM, USB])
def test_shim_inner_boundary_reproduces_oracle(oracle, mode):
    _build_demo()
    x = synth.lcg_bytes(13, 4 * 32768)
    out = subprocess.run([DEMO, str(mode), "inner", "32768"], input=x.tobytes(), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, check=True).stdout
    got = np.frombuffer(out, dtype=np.int16)
    o = oracle.demod(mode)
    want = np.concatenate([o.process(x[b * 32768:(b + 1) * 32768]) for b in range(4)])
    assert (got == want).all()


@pytest.mark.gpu
def test_shim_ssb_modulator_reproduces_oracle(oracle):
    _build_demo()
    pcm = synth.lcg_pcm(7, 3 * 512)
    out = subprocess.run([DEMO, "5", "ssbmod", "0"], input=pcm.tobytes(), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, check=True).stdout
    got = np.frombuffer(out, dtype=np.int8)
    o = oracle.ssbmod(False)
    want = np.concatenate([o.process(pcm[b * 512:(b + 1) * 512]) for b in range(3)])
    assert (got == want).all()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,arg,param,tol", [("ammod", "500", 0.5, 0), ("fmmod", "1200", 1200.0, 0),
                                                ("wbfmmod", "30000", 30000.0, 0)])
def test_shim_am_fm_modulators_reproduce_oracle(oracle, kind, arg, param, tol):
    """AmModulator / FmModulator / WbFmModulator shim classes (reference names and setters) against
    the oracle: bit-exact, all three (FM since round 5: glibc's cosf / sinf restated on the device)"""
    _build_demo()
    pcm = synth.lcg_pcm(9, 2 * 512)
    out = subprocess.run([DEMO, arg, kind, "0"], input=pcm.tobytes(), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, check=True).stdout
    got = np.frombuffer(out, dtype=np.int8)
    o = getattr(oracle, kind)()
    o.set_param(param)
    want = np.concatenate([o.process(pcm[b * 512:(b + 1) * 512]) for b in range(2)])
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    d = np.minimum(d, 256 - d)
    assert len(got) == len(want) and d.max() <= tol


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4, 5])
def test_shim_baseband_data_processor_dispatch(oracle, mode):
    """BasebandDataProcessor::getIqData -> modulateBasebandData (BasebandDataProcessor.cc:381,
    630-697) through the shim class: one 512-sample block off the PCM ring (drop / repeat pacing, zeros while the
    stream is idle) through the modulator of the mode; mode None fills the transfer buffer with 64.  Expected: the
    oracle's ring model driven with the same schedule feeding the oracle's modulator.  Bit-exact in every mode (FM, mode 2,
    since round 5: FmModulator goes through libm cosf/sinf in the reference, glibc's algorithm restated on the device)."""
    _build_demo()
    ops = "r" + "w" * 16 + "s" + "r" * 3 + "wr" * 6 + "rrrr" + "w" * 9 + "rr" + "p" + "r"
    n_w = ops.count("w")
    pcm = synth.lcg_pcm(21, n_w * 512)
    out = subprocess.run([DEMO, str(mode), "bbp", "0", ops], input=pcm.tobytes(), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, check=True).stdout
    got = np.frombuffer(out, dtype=np.int8).reshape(-1, 262144)
    ring = oracle.txring()
    mod = {1: lambda: oracle.ammod(), 2: lambda: oracle.fmmod(), 3: lambda: oracle.wbfmmod(), 4: lambda: oracle.ssbmod(True),
           5: lambda: oracle.ssbmod(False)}.get(mode, lambda: None)()
    want, w = [], 0
    for o in ops:
        if o == "w":
            ring.write(pcm[w * 512:(w + 1) * 512]); w += 1
        elif o == "r":
            blk = ring.read()
            want.append(np.full(262144, 64, dtype=np.int8) if mod is None else mod.process(blk))
        else:
            ring.set_running(o == "s")
    want = np.stack(want)
    assert got.shape == want.shape
    assert (got == want).all()                             # every mode, FM (2) included since round 5
    st = ring.stats()
    assert st[2] > 0 and st[3] > 0, "the schedule should exercise both the drop and the repeat branch"


@pytest.mark.parametrize("up", [1, 0])
def test_shim_fs4_helpers(up):
    """IqDataProcessor::upconvertByFsOver4 / downconvertByFsOver4 (IqDataProcessor.cc:700-815):
    sample n times j^n / (-j)^n with wrapping int8 negation; down undoes up exactly"""
    _build_demo()
    x = synth.lcg_bytes(5, 4096).view(np.int8).copy()
    x[:8] = [-128, -128, -128, 127, 127, -128, 0, -128]
    out = subprocess.run([DEMO, str(up), "fs4", "4096"], input=x.tobytes(), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, check=True).stdout
    got = np.frombuffer(out, dtype=np.int8)
    i, q = x[0::2].astype(np.int16), x[1::2].astype(np.int16)
    n = np.arange(len(i)) % 4
    if up:     # 0 (I,Q)  1 (-Q,I)  2 (-I,-Q)  3 (Q,-I)
        wi = np.select([n == 0, n == 1, n == 2], [i, -q, -i], q)
        wq = np.select([n == 0, n == 1, n == 2], [q, i, -q], -i)
    else:      # 0 (I,Q)  1 (Q,-I)  2 (-I,-Q)  3 (-Q,I)
        wi = np.select([n == 0, n == 1, n == 2], [i, q, -i], -q)
        wq = np.select([n == 0, n == 1, n == 2], [q, -i, -q], i)
    want = np.empty_like(x)
    want[0::2] = wi.astype(np.int8)
    want[1::2] = wq.astype(np.int8)
    assert (got == want).all()
