"""Tooling either side of the hot path, CPU part: the oracle's restatement
of signals/{am,dsb,pm,fm}.cc against the reference-generated fixtures (and the reference programs
themselves where oracle/_ref exists), and the `enable iqdump` wire format as the shim emits it (through a UdpClient: the application's own, a test double here) against
the reference's datagram sequence."""
import json
import os
import subprocess

import numpy as np
import pytest

from hackrfdiags_amd import synth
from tests import reflib, toolsupport as T

HERE = os.path.dirname(os.path.abspath(__file__))
ARR = np.load(os.path.join(HERE, "golden", "golden_tools.npz"))
MAN = json.load(open(os.path.join(HERE, "golden", "golden_tools.json")))


@pytest.mark.parametrize("case", MAN["siggen"], ids=lambda c: c["kind"])
def test_oracle_siggen_reproduces_reference_fixture(oracle, case):
    pcm = synth.lcg_pcm(case["seed"], case["n"])
    pairs, _ = T.orc_siggen(oracle, case["kind"], pcm)
    assert (pairs == ARR[f"sig_{case['kind']}_pairs"]).all()
    iq = oracle.interp().process(pairs)                   # ... | interpolateSignal
    assert iq.size == case["iq_bytes"]
    assert (iq[:4096] == ARR[f"sig_{case['kind']}_iq_head"]).all()
    assert (iq[-4096:] == ARR[f"sig_{case['kind']}_iq_tail"]).all()
    assert synth.digest(iq) == case["iq_sha256"]


@pytest.mark.skipif(not T.have_ref_tools(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("kind", T.SIG_KINDS)
def test_oracle_siggen_equals_reference_program(oracle, kind):
    # full-scale, silence and random PCM; fm's phase wraps many times over 20000 samples
    pcm = np.concatenate([synth.lcg_pcm(91, 20000), np.full(300, 32767, np.int16), np.full(300, -32768, np.int16),
                          np.zeros(50, np.int16)])
    pairs, _ = T.orc_siggen(oracle, kind, pcm)
    assert (pairs == T.ref_siggen(kind, pcm)).all()
    # the phase carries across calls like the program's variable across its loop
    a, th = T.orc_siggen(oracle, kind, pcm[:777])
    b, _ = T.orc_siggen(oracle, kind, pcm[777:], th)
    assert (np.concatenate([a, b]) == pairs).all()


@pytest.mark.parametrize("case", MAN["udp"], ids=lambda c: str(c["bytes"]))
def test_shim_udpclient_datagrams(case):
    """2048-byte datagrams and a remainder (UdpClient.cc:173-241), payload = the raw bytes."""
    exe = T.build_udp_demo()
    cap = T.UdpCapture()
    data = synth.lcg_bytes(7, case["bytes"])
    subprocess.run([exe, str(cap.port), str(case["bytes"])], input=data.tobytes(), check=True)
    got = cap.drain()
    cap.close()
    assert [len(g) for g in got] == case["datagrams"]
    assert b"".join(got) == data.tobytes()


@pytest.mark.skipif(not reflib.have_ref(), reason="oracle/_ref not built")
def test_shim_udpclient_equals_reference_udpclient():
    import ctypes as C
    ref = reflib.Ref()
    ref.lib.ref_udp_send.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int]
    exe = T.build_udp_demo()
    for n in (32768, 4097, 1):
        data = synth.lcg_bytes(11, n)
        cap = T.UdpCapture()
        ref.lib.ref_udp_send(b"127.0.0.1", cap.port, data.ctypes.data, n)
        want = cap.drain()
        subprocess.run([exe, str(cap.port), str(n)], input=data.tobytes(), check=True)
        got = cap.drain()
        cap.close()
        assert got == want and len(got) > 0


@pytest.mark.skipif(not reflib.have_ref(), reason="oracle/_ref not built")
def test_playback_model_equals_reference_dataprovider(tmp_path):
    """tests/toolsupport.playback_model (what the GPU test checks hrfd_play against) IS
    DataProvider::getIqData (DataProvider.cc:174-231): odd file length, many wraps."""
    import ctypes as C
    ref = reflib.Ref()
    L = ref.lib
    L.ref_provider_create.restype = C.c_void_p
    L.ref_provider_destroy.argtypes = [C.c_void_p]
    L.ref_provider_load.argtypes = [C.c_void_p, C.c_char_p]
    L.ref_provider_get.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    image = synth.lcg_bytes(3, 100003)
    path = tmp_path / "x.iq"
    image.tofile(path)
    p = C.c_void_p(L.ref_provider_create())
    assert L.ref_provider_load(p, str(path).encode()) == 1
    idx = 0
    for n in (262144, 5, 99999, 262144, 100003, 1):
        out = np.zeros(n, dtype=np.int8)
        L.ref_provider_get(p, out.ctypes.data, n)
        want, idx = T.playback_model(image, idx, n)
        assert (out == want).all()
    L.ref_provider_destroy(p)
