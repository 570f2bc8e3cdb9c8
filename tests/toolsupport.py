"""Helpers of the tooling tests (UDP capture, reference generator binaries).  Test infrastructure."""
from __future__ import annotations

import os
import socket
import subprocess

import numpy as np

from . import reflib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "hackrfdiags_amd", "csrc", "shim")
SIG_KINDS = ["am", "dsb", "pm", "fm"]          # orc_siggen kind 0..3


class UdpCapture:
    """A bound datagram socket on 127.0.0.1; `drain()` returns every datagram received so far."""

    def __init__(self):
        self.s = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        self.s.setsockopt(socket.SOL_SOCKET, socket.SO_RCVBUF, 1 << 22)
        self.s.bind(("127.0.0.1", 0))
        self.port = self.s.getsockname()[1]
        self.s.settimeout(0.5)

    def drain(self):
        out = []
        while True:
            try:
                out.append(self.s.recv(65536))
            except socket.timeout:
                return out

    def close(self):
        self.s.close()


def ref_sig_binary(kind: str) -> str:
    return os.path.join(reflib.ORACLE_DIR, "_ref", "sig_" + kind)


def have_ref_tools() -> bool:
    return all(os.path.exists(ref_sig_binary(k)) for k in SIG_KINDS) and os.path.exists(reflib.REF_INTERP)


def ref_siggen(kind: str, pcm: np.ndarray) -> np.ndarray:
    """The reference's own program: int16 PCM on stdin -> int16 (I,Q) pairs on stdout."""
    out = subprocess.run([ref_sig_binary(kind)], input=np.ascontiguousarray(pcm, dtype=np.int16).tobytes(),
                         stdout=subprocess.PIPE, check=True).stdout
    return np.frombuffer(out, dtype=np.int16).copy()


def ref_interpolate(pairs: np.ndarray) -> np.ndarray:
    out = subprocess.run([reflib.REF_INTERP], input=np.ascontiguousarray(pairs, dtype=np.int16).tobytes(),
                         stdout=subprocess.PIPE, check=True).stdout
    return np.frombuffer(out, dtype=np.int8).copy()


def orc_siggen(oracle, kind: str, pcm: np.ndarray, theta: float = 0.0):
    import ctypes as C
    L = oracle.lib
    L.orc_siggen.argtypes = [C.c_int, C.POINTER(C.c_int16), C.c_uint32, C.POINTER(C.c_int16), C.POINTER(C.c_float)]
    L.orc_siggen.restype = None
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    out = np.zeros(2 * len(pcm), dtype=np.int16)
    th = C.c_float(theta)
    L.orc_siggen(SIG_KINDS.index(kind), pcm.ctypes.data_as(C.POINTER(C.c_int16)), len(pcm),
                 out.ctypes.data_as(C.POINTER(C.c_int16)), C.byref(th))
    return out, th.value


def build_udp_demo() -> str:
    exe = os.path.join(ROOT, "tests", "cpp", "udp_demo")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "cpp", "udp_demo.cc"),
                           "-I", os.path.join(ROOT, "tests", "cpp")])
    return exe


def playback_model(image: np.ndarray, index: int, nbytes: int):
    """DataProvider::retrieveIqDataFromBuffer as index arithmetic."""
    idx = (index + np.arange(nbytes, dtype=np.int64)) % len(image)
    return image[idx], int((index + nbytes) % len(image))


def retune_minus_64k(iq: np.ndarray) -> np.ndarray:
    """The channel between the reference's transmitter and its receiver: int8 IQ at 2.048 MS/s moved DOWN by 64 kHz.
    The radio tunes its receiver 64 kHz (a quarter of the 256 kS/s rate) high and IqDataProcessor::upconvertByFsOver4
    brings the signal back to 0 Hz (Radio.cc:1187-1191, IqDataProcessor.cc:771-815): a transmitter's baseband therefore
    reaches the demodulators only through this shift.  exp(-j pi n / 16), rounded to the nearest integer, clipped to
    int8.  Test infrastructure (numpy, float64); the golden manifest holds the sha256 of what it produced there."""
    x = np.ascontiguousarray(iq, dtype=np.int8).astype(np.float64)
    z = x[0::2] + 1j * x[1::2]
    n = np.arange(z.size, dtype=np.int64) % 32
    z = z * np.exp(-1j * np.pi * n / 16.0)
    out = np.empty(iq.size, dtype=np.int8)
    out[0::2] = np.clip(np.rint(z.real), -128, 127).astype(np.int8)
    out[1::2] = np.clip(np.rint(z.imag), -128, 127).astype(np.int8)
    return out
