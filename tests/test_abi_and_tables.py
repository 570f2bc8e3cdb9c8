"""CPU-side checks of the product: the C-ABI library loads without a GPU,
exports every symbol include/hrfd.h declares, refuses to work without a device
(no CPU fallback), and carries the right constant tables."""
import ctypes
import os
import re

import numpy as np
import pytest

from hackrfdiags_amd import _lib, api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "hackrfdiags_amd", "csrc")])
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hrfd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hrfd_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/hrfd.h but not exported: {missing}"


def test_no_cpu_fallback(lib):
    if lib.hrfd_device_count() > 0:
        pytest.skip("a GPU is present")
    h = ctypes.c_void_p()
    rc = lib.hrfd_rx_create(4, -1, ctypes.byref(h))
    assert rc == -2 and not h.value                     # HRFD_ENODEV
    assert b"no HIP device" in lib.hrfd_last_error()
    with pytest.raises(api.HrfdError):
        api.Rx(1)


def test_q15_tables_match_oracle_design_values(lib, oracle):
    for name in ["HB1", "HB2", "HB3", "WBFM_D1", "POST_D12", "AUDIO_D40", "FM_TUNER_D32", "AM_D1", "AM_D2",
                 "AM_D3", "SSB_DELAY", "SSB_HILBERT", "INTERP_HB8", "INTERP_HB3", "INTERP_HB2", "INTERP_HB1",
                 "INTERPSIG_S1"]:
        assert (api.q15_table(name) == oracle.quantise(oracle.table(name))).all(), name
    assert len(api.q15_table("no such table")) == 0


def test_host_built_tables_match_oracle(lib, oracle):
    a = api.atan2_table().view(np.uint32)
    b = oracle.atan2_lut().view(np.uint32)
    assert (a == b).all()
    assert (api.dbfs_table() == oracle.dbfs_table()).all()


def test_halfband_offset_domain_formula():
    """The packed 16-bit form the kernel uses for the three half-band stages
    (hrfd_rx_kernels.hip, halfband<D,K,SH>) equals the reference's Q15 form for
    every reachable input, and never leaves the int16 range."""
    for h0, d, k, sh, rng in [(8206, 14, 12800, 13, 129), (8249, 57, 1792, 13, 130), (8424, 29, -5376, 10, 132)]:
        s = np.arange(-2 * rng, 2 * rng + 1)[:, None]      # a + c
        b = np.arange(-rng, rng + 1)[None, :]
        direct = (16384 + h0 * s + 16384 * b) >> 15
        t = s + 256
        kk = t * d + k
        assert kk.min() >= -32768 and kk.max() <= 32767
        u = t + 2 * (b + 128) + (kk >> sh)
        assert u.min() >= -32768 and u.max() <= 32767
        assert ((u >> 2) - 128 == direct).all(), h0
