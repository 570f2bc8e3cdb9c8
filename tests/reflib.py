"""ctypes access to the two CHECKERS: oracle/libhrfd_oracle.so (our CPU
restatement) and oracle/_ref/libhrfd_ref.so (the reference's own sources,
compiled by oracle/Makefile).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libhrfd_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libhrfd_ref.so")
REF_INTERP = os.path.join(ORACLE_DIR, "_ref", "interpolateSignal")

NONE, AM, FM, WBFM, LSB, USB = range(6)
MODE_NAMES = {AM: "am", FM: "fm", WBFM: "wbfm", LSB: "lsb", USB: "usb"}

_i8p = C.POINTER(C.c_int8)
_i16p = C.POINTER(C.c_int16)
_u32p = C.POINTER(C.c_uint32)
_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)


def _p(a, t):
    return a.ctypes.data_as(t)


def iq256_capacity(block_bytes: int) -> int:
    """bytes of the 256 kS/s stream a call of block_bytes can complete at most: 2 * ceil(block_bytes / 16)"""
    return 2 * ((int(block_bytes) // 2 + 7) // 8)


def build_oracle():
    if not os.path.exists(ORACLE_SO) or (
            os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(ORACLE_DIR, "hrfd_oracle.c"))):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)


def have_ref() -> bool:
    return os.path.exists(REF_SO)


class Oracle:
    """Our CPU restatement."""
    prefix = "orc"

    def __init__(self):
        build_oracle()
        self.lib = L = C.CDLL(ORACLE_SO)
        L.orc_rx_create.restype = C.c_void_p
        L.orc_rx_destroy.argtypes = [C.c_void_p]
        L.orc_rx_set_mode.argtypes = [C.c_void_p, C.c_int]
        L.orc_rx_set_gain.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.orc_rx_set_threshold.argtypes = [C.c_void_p, C.c_int32]
        L.orc_rx_process.restype = C.c_uint32
        L.orc_rx_process.argtypes = [C.c_void_p, _i8p, C.c_uint32, C.c_uint32, _i16p, C.c_uint32,
                                     _u32p, C.POINTER(C.c_int), _i8p]
        L.orc_rx_wbfm_float_stream.restype = C.c_uint32
        L.orc_rx_wbfm_float_stream.argtypes = [C.c_void_p, _f32p, C.c_uint32]
        L.orc_demod_create.restype = C.c_void_p
        L.orc_demod_create.argtypes = [C.c_int]
        L.orc_demod_destroy.argtypes = [C.c_void_p]
        L.orc_demod_reset.argtypes = [C.c_void_p]
        L.orc_demod_set_gain.argtypes = [C.c_void_p, C.c_float]
        L.orc_demod_set_sideband.argtypes = [C.c_void_p, C.c_int]
        L.orc_demod_process.restype = C.c_uint32
        L.orc_demod_process.argtypes = [C.c_void_p, _i8p, C.c_uint32, _i16p, C.c_uint32]
        for kind, setter in (("ammod", "set_index"), ("fmmod", "set_deviation"), ("wbfmmod", "set_deviation")):
            getattr(L, f"orc_{kind}_create").restype = C.c_void_p
            getattr(L, f"orc_{kind}_create").argtypes = []
            getattr(L, f"orc_{kind}_destroy").argtypes = [C.c_void_p]
            getattr(L, f"orc_{kind}_reset").argtypes = [C.c_void_p]
            getattr(L, f"orc_{kind}_{setter}").argtypes = [C.c_void_p, C.c_float]
            getattr(L, f"orc_{kind}_process").restype = C.c_uint32
            getattr(L, f"orc_{kind}_process").argtypes = [C.c_void_p, _i16p, C.c_uint32, _i8p]
        L.orc_txring_create.restype = C.c_void_p
        L.orc_txring_create.argtypes = []
        L.orc_txring_destroy.argtypes = [C.c_void_p]
        L.orc_txring_set_running.argtypes = [C.c_void_p, C.c_int]
        L.orc_txring_write.argtypes = [C.c_void_p, _i16p]
        L.orc_txring_read.argtypes = [C.c_void_p, _i16p]
        L.orc_txring_stats.argtypes = [C.c_void_p, _u32p]
        L.orc_ssbmod_create.restype = C.c_void_p
        L.orc_ssbmod_create.argtypes = [C.c_int]
        L.orc_ssbmod_destroy.argtypes = [C.c_void_p]
        L.orc_ssbmod_reset.argtypes = [C.c_void_p]
        L.orc_ssbmod_set_sideband.argtypes = [C.c_void_p, C.c_int]
        L.orc_ssbmod_process.restype = C.c_uint32
        L.orc_ssbmod_process.argtypes = [C.c_void_p, _i16p, C.c_uint32, _i8p]
        L.orc_interp_create.restype = C.c_void_p
        L.orc_interp_destroy.argtypes = [C.c_void_p]
        L.orc_interp_process.restype = C.c_uint32
        L.orc_interp_process.argtypes = [C.c_void_p, _i16p, C.c_uint32, _i8p]
        L.orc_nco_create.restype = C.c_void_p
        L.orc_nco_create.argtypes = [C.c_float, C.c_float]
        L.orc_nco_destroy.argtypes = [C.c_void_p]
        L.orc_nco_set_frequency.argtypes = [C.c_void_p, C.c_float]
        L.orc_nco_reset.argtypes = [C.c_void_p]
        L.orc_nco_run.argtypes = [C.c_void_p, C.c_int, C.c_uint32, _f32p, _f32p]
        L.orc_nco_tables.argtypes = [C.c_void_p, _f32p, _f32p]
        L.orc_quantise.argtypes = [_f32p, C.c_int, _i16p]
        L.orc_decimate.restype = C.c_uint32
        L.orc_decimate.argtypes = [_f32p, C.c_int, C.c_int, _i16p, C.c_uint32, _i16p]
        L.orc_interpolate.argtypes = [_f32p, C.c_int, C.c_int, _i16p, C.c_uint32, _i16p]
        L.orc_iir.argtypes = [_f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_uint32, _f32p]
        L.orc_fir.argtypes = [_f32p, C.c_int, _f32p, C.c_uint32, _f32p]
        L.orc_float_to_int16.restype = C.c_int16
        L.orc_float_to_int16.argtypes = [C.c_float]
        L.orc_atan2_lut.argtypes = [_f32p]
        L.orc_dbfs_table.argtypes = [_i32p]
        L.orc_table.restype = C.c_int
        L.orc_table.argtypes = [C.c_char_p, _f32p, C.c_int]

    # ---- tables / primitives
    def table(self, name: str) -> np.ndarray:
        buf = np.zeros(64, dtype=np.float32)
        n = self.lib.orc_table(name.encode(), _p(buf, _f32p), 64)
        assert n > 0, name
        return buf[:n].copy()

    def quantise(self, coeffs) -> np.ndarray:
        c = np.ascontiguousarray(coeffs, dtype=np.float32)
        out = np.zeros(len(c), dtype=np.int16)
        self.lib.orc_quantise(_p(c, _f32p), len(c), _p(out, _i16p))
        return out

    def decimate(self, coeffs, factor, x) -> np.ndarray:
        c = np.ascontiguousarray(coeffs, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.int16)
        out = np.zeros(len(x) // factor + 2, dtype=np.int16)
        n = self.lib.orc_decimate(_p(c, _f32p), len(c), factor, _p(x, _i16p), len(x), _p(out, _i16p))
        return out[:n].copy()

    def interpolate(self, coeffs, factor, x) -> np.ndarray:
        c = np.ascontiguousarray(coeffs, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.int16)
        out = np.zeros(len(x) * factor, dtype=np.int16)
        self.lib.orc_interpolate(_p(c, _f32p), len(c), factor, _p(x, _i16p), len(x), _p(out, _i16p))
        return out

    def fir(self, h, x) -> np.ndarray:
        h = np.ascontiguousarray(h, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.zeros(len(x), dtype=np.float32)
        self.lib.orc_fir(_p(h, _f32p), len(h), _p(x, _f32p), len(x), _p(out, _f32p))
        return out

    def iir(self, b, a, x) -> np.ndarray:
        b = np.ascontiguousarray(b, dtype=np.float32)
        a = np.ascontiguousarray(a, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.zeros(len(x), dtype=np.float32)
        self.lib.orc_iir(_p(b, _f32p), len(b), _p(a, _f32p), len(a), _p(x, _f32p), len(x), _p(out, _f32p))
        return out

    def float_to_int16(self, v: float) -> int:
        return int(self.lib.orc_float_to_int16(C.c_float(v)))

    def atan2_lut(self) -> np.ndarray:
        out = np.zeros((256, 256), dtype=np.float32)
        self.lib.orc_atan2_lut(_p(out, _f32p))
        return out

    def dbfs_table(self) -> np.ndarray:
        out = np.zeros(257, dtype=np.int32)
        self.lib.orc_dbfs_table(_p(out, _i32p))
        return out

    # ---- objects
    def rx(self):
        return _OrcRx(self.lib)

    def demod(self, mode):
        return _OrcDemod(self.lib, mode)

    def ssbmod(self, lsb=True):
        return _OrcSsbMod(self.lib, lsb)

    def txring(self):
        return _TxRing(self.lib, "orc")

    def ammod(self):
        return _Mod(self.lib, "orc", "ammod", "set_index")

    def fmmod(self):
        return _Mod(self.lib, "orc", "fmmod", "set_deviation")

    def wbfmmod(self):
        return _Mod(self.lib, "orc", "wbfmmod", "set_deviation")

    def interp(self):
        return _OrcInterp(self.lib)

    def nco(self, fs, f):
        return _Nco(self.lib, "orc", fs, f)


class _OrcRx:
    def __init__(self, lib):
        self.lib = lib
        self.h = C.c_void_p(lib.orc_rx_create())
        self.gain_db = 0
        self.pending = 0                                   # IQ samples the front end's decimators hold back (0..7)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_rx_destroy(self.h)
            self.h = None

    def set_mode(self, mode):
        self.lib.orc_rx_set_mode(self.h, mode)

    def set_gain(self, mode, gain):
        self.lib.orc_rx_set_gain(self.h, mode, C.c_float(gain))

    def set_threshold(self, t):
        self.lib.orc_rx_set_threshold(self.h, t)

    def process(self, iq: np.ndarray):
        """-> (pcm int16[n], magnitude, allowed, iq256 int8[the call's decimatedByteCount: bytes/8 for multiples of 16])"""
        iq = np.ascontiguousarray(iq, dtype=np.int8)
        pcm = np.zeros(len(iq) // 512 + 8, dtype=np.int16)
        mag = C.c_uint32(0)
        allowed = C.c_int(0)
        iq256 = np.zeros(iq256_capacity(len(iq)) + 16, dtype=np.int8)
        n = self.lib.orc_rx_process(self.h, _p(iq, _i8p), len(iq), self.gain_db, _p(pcm, _i16p), len(pcm),
                                    C.byref(mag), C.byref(allowed), _p(iq256, _i8p))
        count = 2 * ((self.pending + len(iq) // 2) // 8)
        self.pending = (self.pending + len(iq) // 2) % 8
        return pcm[:n].copy(), int(mag.value), bool(allowed.value), iq256[:count]

    def wbfm_float_stream(self, count):
        out = np.zeros(count, dtype=np.float32)
        n = self.lib.orc_rx_wbfm_float_stream(self.h, _p(out, _f32p), count)
        return out[:n]


class _OrcDemod:
    def __init__(self, lib, mode):
        self.lib = lib
        self.h = C.c_void_p(lib.orc_demod_create(mode))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_demod_destroy(self.h)
            self.h = None

    def reset(self):
        self.lib.orc_demod_reset(self.h)

    def set_gain(self, g):
        self.lib.orc_demod_set_gain(self.h, C.c_float(g))

    def set_sideband(self, lsb):
        self.lib.orc_demod_set_sideband(self.h, int(bool(lsb)))

    def process(self, iq256):
        iq256 = np.ascontiguousarray(iq256, dtype=np.int8)
        pcm = np.zeros(len(iq256) // 2 + 8, dtype=np.int16)
        n = self.lib.orc_demod_process(self.h, _p(iq256, _i8p), len(iq256), _p(pcm, _i16p), len(pcm))
        return pcm[:n].copy()


class _TxRing:
    """BasebandDataProcessor's PCM ring in either library (one channel)."""

    def __init__(self, lib, prefix):
        self.lib, self.pre = lib, prefix
        self.h = C.c_void_p(getattr(lib, f"{prefix}_txring_create")())

    def __del__(self):
        if getattr(self, "h", None):
            getattr(self.lib, f"{self.pre}_txring_destroy")(self.h)
            self.h = None

    def set_running(self, running):
        getattr(self.lib, f"{self.pre}_txring_set_running")(self.h, int(bool(running)))

    def write(self, pcm512):
        pcm512 = np.ascontiguousarray(pcm512, dtype=np.int16)
        getattr(self.lib, f"{self.pre}_txring_write")(self.h, _p(pcm512, _i16p))

    def read(self):
        out = np.zeros(512, dtype=np.int16)
        getattr(self.lib, f"{self.pre}_txring_read")(self.h, _p(out, _i16p))
        return out

    def stats(self):
        out = np.zeros(6, dtype=np.uint32)
        getattr(self.lib, f"{self.pre}_txring_stats")(self.h, _p(out, _u32p))
        return out


class _Mod:
    """AM / FM modulator of either library (`orc_` restatement or `ref_` compiled reference).
    The reference's objects take at most 512 PCM samples per call (fixed member arrays)."""

    def __init__(self, lib, prefix, kind, setter, max_call=None):
        self.lib, self.pre, self.kind, self.setter, self.max_call = lib, prefix, kind, setter, max_call
        self.h = C.c_void_p(getattr(lib, f"{prefix}_{kind}_create")())

    def __del__(self):
        if getattr(self, "h", None):
            getattr(self.lib, f"{self.pre}_{self.kind}_destroy")(self.h)
            self.h = None

    def reset(self):
        getattr(self.lib, f"{self.pre}_{self.kind}_reset")(self.h)

    def set_param(self, value):
        getattr(self.lib, f"{self.pre}_{self.kind}_{self.setter}")(self.h, float(value))

    def process(self, pcm):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        step = self.max_call or max(len(pcm), 1)
        outs = []
        for o in range(0, len(pcm), step):
            part = np.ascontiguousarray(pcm[o:o + step])
            out = np.zeros(len(part) * 512, dtype=np.int8)
            n = getattr(self.lib, f"{self.pre}_{self.kind}_process")(self.h, _p(part, _i16p), len(part), _p(out, _i8p))
            outs.append(out[:n])
        return np.concatenate(outs) if outs else np.zeros(0, dtype=np.int8)


class _OrcSsbMod:
    def __init__(self, lib, lsb):
        self.lib = lib
        self.h = C.c_void_p(lib.orc_ssbmod_create(int(bool(lsb))))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_ssbmod_destroy(self.h)
            self.h = None

    def reset(self):
        self.lib.orc_ssbmod_reset(self.h)

    def set_sideband(self, lsb):
        self.lib.orc_ssbmod_set_sideband(self.h, int(bool(lsb)))

    def process(self, pcm):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        out = np.zeros(len(pcm) * 512, dtype=np.int8)
        n = self.lib.orc_ssbmod_process(self.h, _p(pcm, _i16p), len(pcm), _p(out, _i8p))
        return out[:n]


class _OrcInterp:
    def __init__(self, lib):
        self.lib = lib
        self.h = C.c_void_p(lib.orc_interp_create())

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_interp_destroy(self.h)
            self.h = None

    def process(self, iq16):
        iq16 = np.ascontiguousarray(iq16, dtype=np.int16)
        n_pairs = len(iq16) // 2
        out = np.zeros(n_pairs * 512, dtype=np.int8)
        self.lib.orc_interp_process(self.h, _p(iq16, _i16p), n_pairs, _p(out, _i8p))
        return out


class _Nco:
    def __init__(self, lib, prefix, fs, f):
        self.lib, self.px = lib, prefix
        fn = getattr(lib, f"{prefix}_nco_create")
        fn.restype = C.c_void_p
        fn.argtypes = [C.c_float, C.c_float]
        self.h = C.c_void_p(fn(C.c_float(fs), C.c_float(f)))
        for nm, at in (("destroy", [C.c_void_p]), ("set_frequency", [C.c_void_p, C.c_float]),
                       ("reset", [C.c_void_p]), ("run", [C.c_void_p, C.c_int, C.c_uint32, _f32p, _f32p]),
                       ("tables", [C.c_void_p, _f32p, _f32p])):
            getattr(lib, f"{prefix}_nco_{nm}").argtypes = at

    def __del__(self):
        if getattr(self, "h", None):
            getattr(self.lib, f"{self.px}_nco_destroy")(self.h)
            self.h = None

    def set_frequency(self, f):
        getattr(self.lib, f"{self.px}_nco_set_frequency")(self.h, C.c_float(f))

    def reset(self):
        getattr(self.lib, f"{self.px}_nco_reset")(self.h)

    def run(self, n, fast=False):
        i = np.zeros(n, dtype=np.float32)
        q = np.zeros(n, dtype=np.float32)
        getattr(self.lib, f"{self.px}_nco_run")(self.h, int(fast), n, _p(i, _f32p), _p(q, _f32p))
        return i, q

    def tables(self):
        s = np.zeros(16384, dtype=np.float32)
        c = np.zeros(16384, dtype=np.float32)
        getattr(self.lib, f"{self.px}_nco_tables")(self.h, _p(s, _f32p), _p(c, _f32p))
        return s, c


class Ref:
    """The reference's own compiled sources (oracle/_ref)."""

    def __init__(self):
        if not have_ref():
            raise FileNotFoundError(REF_SO)
        self.lib = L = C.CDLL(REF_SO)
        L.ref_rx_create.restype = C.c_void_p
        L.ref_rx_destroy.argtypes = [C.c_void_p]
        L.ref_rx_set_mode.argtypes = [C.c_void_p, C.c_int]
        L.ref_rx_set_gain.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.ref_rx_set_threshold.argtypes = [C.c_void_p, C.c_int32]
        L.ref_set_receive_gain_db.argtypes = [C.c_uint32]
        L.ref_rx_process.restype = C.c_uint32
        L.ref_rx_process.argtypes = [C.c_void_p, _i8p, C.c_uint32, _i16p, C.c_uint32, _u32p, _i8p]
        L.ref_rx_wbfm_float_stream.argtypes = [C.c_void_p, _f32p, C.c_uint32]
        L.ref_rx_reduce_sample_rate.restype = C.c_uint32
        L.ref_rx_reduce_sample_rate.argtypes = [C.c_void_p, _i8p, C.c_uint32, _i8p]
        L.ref_demod_create.restype = C.c_void_p
        L.ref_demod_create.argtypes = [C.c_int]
        L.ref_demod_destroy.argtypes = [C.c_void_p]
        L.ref_demod_reset.argtypes = [C.c_void_p]
        L.ref_demod_set_gain.argtypes = [C.c_void_p, C.c_float]
        L.ref_demod_set_sideband.argtypes = [C.c_void_p, C.c_int]
        L.ref_demod_process.restype = C.c_uint32
        L.ref_demod_process.argtypes = [C.c_void_p, _i8p, C.c_uint32, _i16p, C.c_uint32]
        for kind, setter in (("ammod", "set_index"), ("fmmod", "set_deviation"), ("wbfmmod", "set_deviation")):
            getattr(L, f"ref_{kind}_create").restype = C.c_void_p
            getattr(L, f"ref_{kind}_create").argtypes = []
            getattr(L, f"ref_{kind}_destroy").argtypes = [C.c_void_p]
            getattr(L, f"ref_{kind}_reset").argtypes = [C.c_void_p]
            getattr(L, f"ref_{kind}_{setter}").argtypes = [C.c_void_p, C.c_float]
            getattr(L, f"ref_{kind}_process").restype = C.c_uint32
            getattr(L, f"ref_{kind}_process").argtypes = [C.c_void_p, _i16p, C.c_uint32, _i8p]
        L.ref_txring_create.restype = C.c_void_p
        L.ref_txring_create.argtypes = []
        L.ref_txring_destroy.argtypes = [C.c_void_p]
        L.ref_txring_set_running.argtypes = [C.c_void_p, C.c_int]
        L.ref_txring_write.argtypes = [C.c_void_p, _i16p]
        L.ref_txring_read.argtypes = [C.c_void_p, _i16p]
        L.ref_txring_stats.argtypes = [C.c_void_p, _u32p]
        L.ref_ssbmod_create.restype = C.c_void_p
        L.ref_ssbmod_create.argtypes = [C.c_int]
        L.ref_ssbmod_destroy.argtypes = [C.c_void_p]
        L.ref_ssbmod_reset.argtypes = [C.c_void_p]
        L.ref_ssbmod_set_sideband.argtypes = [C.c_void_p, C.c_int]
        L.ref_ssbmod_process.restype = C.c_uint32
        L.ref_ssbmod_process.argtypes = [C.c_void_p, _i16p, C.c_uint32, _i8p]
        L.ref_quantise.argtypes = [_f32p, C.c_int, _i16p]
        L.ref_decimate.restype = C.c_uint32
        L.ref_decimate.argtypes = [_f32p, C.c_int, C.c_int, _i16p, C.c_uint32, _i16p]
        L.ref_interpolate.argtypes = [_f32p, C.c_int, C.c_int, _i16p, C.c_uint32, _i16p]
        L.ref_iir.argtypes = [_f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_uint32, _f32p]
        L.ref_float_to_int16.restype = C.c_int16
        L.ref_float_to_int16.argtypes = [C.c_float]
        L.ref_dbfs_table.argtypes = [_i32p]
        L.ref_magnitude_to_dbfs.restype = C.c_int32
        L.ref_magnitude_to_dbfs.argtypes = [C.c_uint32]
        L.ref_bench_rx.restype = C.c_uint64
        L.ref_bench_rx.argtypes = [C.c_int, C.c_uint32, C.c_double, _i8p, C.c_uint32, C.c_uint32,
                                   C.POINTER(C.c_double), C.POINTER(C.c_uint64)]

    def bench_rx(self, mode, threads, seconds, iq):
        """bench.py's cpu_baseline leg: `threads` std::threads inside the harness, one IqDataProcessor and its
        demodulators per thread, each looping over iq [n_blocks][block_bytes] for `seconds` (one ctypes call: the
        interpreter is not in the loop).  -> (blocks processed by all threads, wall seconds, PCM samples produced)"""
        iq = np.ascontiguousarray(iq, dtype=np.int8)
        nb, bb = iq.shape
        dt, pcm = C.c_double(0.0), C.c_uint64(0)
        n = self.lib.ref_bench_rx(mode, threads, seconds, _p(iq, _i8p), nb, bb, C.byref(dt), C.byref(pcm))
        return int(n), float(dt.value), int(pcm.value)

    def quantise(self, coeffs):
        c = np.ascontiguousarray(coeffs, dtype=np.float32)
        out = np.zeros(len(c), dtype=np.int16)
        self.lib.ref_quantise(_p(c, _f32p), len(c), _p(out, _i16p))
        return out

    def decimate(self, coeffs, factor, x):
        c = np.ascontiguousarray(coeffs, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.int16)
        out = np.zeros(len(x) // factor + 2, dtype=np.int16)
        n = self.lib.ref_decimate(_p(c, _f32p), len(c), factor, _p(x, _i16p), len(x), _p(out, _i16p))
        return out[:n].copy()

    def interpolate(self, coeffs, factor, x):
        c = np.ascontiguousarray(coeffs, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.int16)
        out = np.zeros(len(x) * factor, dtype=np.int16)
        self.lib.ref_interpolate(_p(c, _f32p), len(c), factor, _p(x, _i16p), len(x), _p(out, _i16p))
        return out

    def iir(self, b, a, x):
        b = np.ascontiguousarray(b, dtype=np.float32)
        a = np.ascontiguousarray(a, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.zeros(len(x), dtype=np.float32)
        self.lib.ref_iir(_p(b, _f32p), len(b), _p(a, _f32p), len(a), _p(x, _f32p), len(x), _p(out, _f32p))
        return out

    def float_to_int16(self, v):
        return int(self.lib.ref_float_to_int16(C.c_float(v)))

    def dbfs_table(self):
        out = np.zeros(257, dtype=np.int32)
        self.lib.ref_dbfs_table(_p(out, _i32p))
        return out

    def rx(self):
        return _RefRx(self.lib)

    def demod(self, mode):
        return _RefDemod(self.lib, mode)

    def ssbmod(self, lsb=True):
        return _RefSsbMod(self.lib, lsb)

    def txring(self):
        return _TxRing(self.lib, "ref")

    def ammod(self):
        return _Mod(self.lib, "ref", "ammod", "set_index", max_call=512)

    def fmmod(self):
        return _Mod(self.lib, "ref", "fmmod", "set_deviation", max_call=512)

    def wbfmmod(self):
        return _Mod(self.lib, "ref", "wbfmmod", "set_deviation", max_call=512)

    def nco(self, fs, f):
        return _Nco(self.lib, "ref", fs, f)

    @staticmethod
    def interpolate_signal(iq16: np.ndarray) -> np.ndarray:
        """Run the reference CLI tool signals/interpolateSignal (stdin -> stdout)."""
        data = np.ascontiguousarray(iq16, dtype=np.int16).tobytes()
        out = subprocess.run([REF_INTERP], input=data, stdout=subprocess.PIPE, check=True).stdout
        return np.frombuffer(out, dtype=np.int8).copy()


class _RefRx:
    def __init__(self, lib):
        self.lib = lib
        self.h = C.c_void_p(lib.ref_rx_create())
        self.gain_db = 0
        self.pending = 0

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_rx_destroy(self.h)
            self.h = None

    def set_mode(self, mode):
        self.lib.ref_rx_set_mode(self.h, mode)

    def set_gain(self, mode, gain):
        self.lib.ref_rx_set_gain(self.h, mode, C.c_float(gain))

    def set_threshold(self, t):
        self.lib.ref_rx_set_threshold(self.h, t)

    def process(self, iq):
        """-> (pcm, magnitude, allowed(None: not observable), iq256)"""
        iq = np.ascontiguousarray(iq, dtype=np.int8)
        pcm = np.zeros(len(iq) // 512 + 8, dtype=np.int16)
        mag = C.c_uint32(0)
        iq256 = np.zeros(iq256_capacity(len(iq)) + 16, dtype=np.int8)
        self.lib.ref_set_receive_gain_db(self.gain_db)
        n = self.lib.ref_rx_process(self.h, _p(iq, _i8p), len(iq), _p(pcm, _i16p), len(pcm),
                                    C.byref(mag), _p(iq256, _i8p))
        count = 2 * ((self.pending + len(iq) // 2) // 8)   # (the count reduceSampleRate returns: test_oracle_vs_ref checks it)
        self.pending = (self.pending + len(iq) // 2) % 8
        return pcm[:n].copy(), int(mag.value), None, iq256[:count]

    def reduce_sample_rate(self, iq):
        """IqDataProcessor::reduceSampleRate alone -> (byteCount it returns, decimatedData[:byteCount], no Fs/4 mix)"""
        iq = np.ascontiguousarray(iq, dtype=np.int8)
        out = np.zeros(iq256_capacity(len(iq)) + 16, dtype=np.int8)
        n = self.lib.ref_rx_reduce_sample_rate(self.h, _p(iq, _i8p), len(iq), _p(out, _i8p))
        self.pending = (self.pending + (len(iq) + 1) // 2) % 8
        return int(n), out[:n].copy()

    def wbfm_float_stream(self, count):
        out = np.zeros(count, dtype=np.float32)
        self.lib.ref_rx_wbfm_float_stream(self.h, _p(out, _f32p), count)
        return out


class _RefDemod(_OrcDemod):
    def __init__(self, lib, mode):
        self.lib = lib
        self.h = C.c_void_p(lib.ref_demod_create(mode))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_demod_destroy(self.h)
            self.h = None

    def reset(self):
        self.lib.ref_demod_reset(self.h)

    def set_gain(self, g):
        self.lib.ref_demod_set_gain(self.h, C.c_float(g))

    def set_sideband(self, lsb):
        self.lib.ref_demod_set_sideband(self.h, int(bool(lsb)))

    def process(self, iq256):
        iq256 = np.ascontiguousarray(iq256, dtype=np.int8)
        pcm = np.zeros(len(iq256) // 2 + 8, dtype=np.int16)
        n = self.lib.ref_demod_process(self.h, _p(iq256, _i8p), len(iq256), _p(pcm, _i16p), len(pcm))
        return pcm[:n].copy()


class _RefSsbMod:
    def __init__(self, lib, lsb):
        self.lib = lib
        self.h = C.c_void_p(lib.ref_ssbmod_create(int(bool(lsb))))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_ssbmod_destroy(self.h)
            self.h = None

    def reset(self):
        self.lib.ref_ssbmod_reset(self.h)

    def set_sideband(self, lsb):
        self.lib.ref_ssbmod_set_sideband(self.h, int(bool(lsb)))

    def process(self, pcm):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        out = np.zeros(len(pcm) * 512, dtype=np.int8)
        n = self.lib.ref_ssbmod_process(self.h, _p(pcm, _i16p), len(pcm), _p(out, _i8p))
        return out[:n]
