#!/usr/bin/env python3
"""Golden vectors for the tooling either side of the hot path, produced by
the REFERENCE's own programs compiled by oracle/Makefile from /root/reference: signals/{am,dsb,pm,fm}.cc
(int16 PCM -> int16 IQ pairs), piped into signals/interpolateSignal.cc (-> int8 IQ at 2.048 MS/s), and
UdpClient::sendData's datagram sizes.  Build container only.

    python tests/golden/make_golden_tools.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from hackrfdiags_amd import synth  # noqa: E402
from tests import reflib, toolsupport as T  # noqa: E402


def main():
    arrays, manifest = {}, {"siggen": [], "udp": []}
    for kind in T.SIG_KINDS:
        pcm = synth.lcg_pcm(31 + T.SIG_KINDS.index(kind), 1024)
        pairs = T.ref_siggen(kind, pcm)
        iq = T.ref_interpolate(pairs)
        arrays[f"sig_{kind}_pairs"] = pairs
        arrays[f"sig_{kind}_iq_head"] = iq[:4096]
        arrays[f"sig_{kind}_iq_tail"] = iq[-4096:]
        manifest["siggen"].append({"kind": kind, "seed": 31 + T.SIG_KINDS.index(kind), "n": 1024,
                                   "iq_sha256": synth.digest(iq), "iq_bytes": int(iq.size)})
    ref = reflib.Ref()
    import ctypes as C
    ref.lib.ref_udp_send.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int]
    for n in (32768, 5000, 2048, 100):
        cap = T.UdpCapture()
        data = synth.lcg_bytes(7, n)
        ref.lib.ref_udp_send(b"127.0.0.1", cap.port, data.ctypes.data, n)
        got = cap.drain()
        cap.close()
        assert b"".join(got) == data.tobytes()
        manifest["udp"].append({"bytes": n, "datagrams": [len(g) for g in got]})
    np.savez_compressed(os.path.join(HERE, "golden_tools.npz"), **arrays)
    with open(os.path.join(HERE, "golden_tools.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("wrote golden_tools.npz / .json:", {k: len(v) for k, v in manifest.items()})


if __name__ == "__main__":
    main()
