"""Tooling either side of the hot path on the GPU: hrfd_play_* (DataProvider's
cyclic .iq playback from HBM), the signals/ generators as hrfd_mod kinds, the `enable iqdump` stream
of the IqDataProcessor shim on its UDP wire format, and the DataProvider shim class."""
import os
import subprocess

import numpy as np
import pytest

from hackrfdiags_amd import api, synth
from tests import toolsupport as T
from tests.reflib import WBFM

pytestmark = pytest.mark.gpu
BLK = synth.BLOCK_BYTES


def test_play_cyclic_playback_many_channels():
    image = synth.lcg_bytes(5, 1000003)                   # odd length: every alignment, wraps mid-call
    C = 5
    p = api.Play(C)
    assert (p.get(64) == 0).all()                         # no file loaded: no action (DataProvider.cc:181)
    p.load(image)
    starts = [0, 1, 999999, 500001, 262144]
    for c, s in enumerate(starts):
        p.set_position(s, c)
    idx = list(starts)
    for n in (BLK, 5, 100001, BLK, 16, 1000003, 3):
        got = p.get(n)
        for c in range(C):
            want, idx[c] = T.playback_model(image, idx[c], n)
            assert (got[c] == want).all(), (n, c)
            assert p.position(c) == idx[c]
    with pytest.raises(api.HrfdError):
        p.set_position(len(image), 0)


def test_play_load_file_and_feed_receiver(oracle, tmp_path):
    """a generated .iq file played cyclically into the WBFM receiver == the oracle on the same bytes"""
    import torch
    image = synth.make_input("fmtone", 7, 3)[: 2 * BLK + 77777]
    path = tmp_path / "fm.iq"
    image.tofile(path)
    C, B = 2, 4
    p = api.Play(C)
    p.load_file(str(path))
    p.set_position(12345, 1)
    dev = torch.device("cuda:0")
    x = torch.zeros((C, B * BLK), dtype=torch.int8, device=dev)
    torch.cuda.synchronize()
    p.get_device(x.data_ptr(), B * BLK, B * BLK)
    torch.cuda.synchronize()
    xs = x.cpu().numpy()
    for c, s in enumerate((0, 12345)):
        want, _ = T.playback_model(image, s, B * BLK)
        assert (xs[c] == want).all()
    rx = api.Rx(C)
    rx.set_mode(api.WBFM)
    pcm = rx.process_block(xs.reshape(C, B, BLK), B)[0]
    for c in range(C):
        o = oracle.rx()
        o.set_mode(WBFM)
        for b in range(B):
            assert (pcm[c, b] == o.process(xs[c, b * BLK:(b + 1) * BLK])[0]).all()


@pytest.mark.parametrize("kind,tol", [("am", 0), ("dsb", 0), ("pm", 0), ("fm", 0)])
def test_signal_generators_match_oracle(oracle, kind, tol):
    """signals/<kind>.cc | interpolateSignal as one hrfd_mod kind: int8 IQ at 2.048 MS/s.  am and dsb
    are integer work; pm and fm evaluate cos/sin (the reference: cosf/sinf -- glibc's algorithm restated on the device since
    round 5).  Bit-exact, all four."""
    C, n, calls = 3, 512, 3
    k = {"am": api.MOD_SIG_AM, "dsb": api.MOD_SIG_DSB, "pm": api.MOD_SIG_PM, "fm": api.MOD_SIG_FM}[kind]
    m = api.Mod(k, C)
    pcm = np.stack([synth.lcg_pcm(60 + c, n * calls) for c in range(C)])
    if kind == "fm":
        pcm[2] = 32767                                    # the phase wraps every few samples
    got = np.concatenate([m.process(pcm[:, i * n:(i + 1) * n]) for i in range(calls)], axis=1)
    worst = 0
    for c in range(C):
        pairs, _ = T.orc_siggen(oracle, kind, pcm[c])
        want = oracle.interp().process(pairs)
        d = np.abs(got[c].astype(np.int32) - want.astype(np.int32))
        worst = max(worst, int(d.max()))
        assert d.max() <= tol, (kind, c, int(d.max()))
        if tol:
            assert (d != 0).mean() < 0.02
    assert tol or worst == 0


def _demo():
    from tests.test_shim import DEMO, _build_demo
    _build_demo()
    return DEMO


def test_shim_iqdump_goes_out_as_udp_datagrams(oracle):
    """IqDataProcessor::enableIqDump through the shim: 16 datagrams of 2048 bytes per block, carrying the
    mixed 256 kS/s stream (IqDataProcessor.cc:953-957, UdpClient.cc:173-241)."""
    demo = _demo()
    B = 3
    x = synth.make_input("fmtone", 9, B)
    cap = T.UdpCapture()
    subprocess.run([demo, "3", "outer", str(BLK), str(cap.port)], input=x.tobytes(), stdout=subprocess.PIPE,
                   stderr=subprocess.PIPE, check=True)
    got = cap.drain()
    cap.close()
    assert [len(g) for g in got] == [2048] * (16 * B)
    o = oracle.rx()
    o.set_mode(WBFM)
    want = np.concatenate([o.process(x[b * BLK:(b + 1) * BLK])[3] for b in range(B)])
    assert np.frombuffer(b"".join(got), dtype=np.int8).tolist() == want.tolist()


def test_shim_dataprovider_class(tmp_path):
    demo = _demo()
    image = synth.lcg_bytes(8, 300001)
    path = tmp_path / "p.iq"
    image.tofile(path)
    out = subprocess.run([demo, "4", "provider", str(BLK), str(path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         check=True).stdout
    want, _ = T.playback_model(image, 0, 4 * BLK)
    assert (np.frombuffer(out, dtype=np.int8) == want).all()
