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


def declared_symbols(header="hrfd.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hrfd_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/hrfd.h but not exported: {missing}"


def test_debug_entry_points_are_declared_and_nothing_else_is_exported(lib):
    """every hrfd_* symbol the shipped library exports is declared in include/hrfd.h (the drop-in boundary) or in
    include/hrfd_debug.h (introspection, and test hooks that are inert without HRFD_DEBUG_HOOKS=1)"""
    import subprocess
    debug = declared_symbols("hrfd_debug.h")
    assert all(("_debug_" in n or n.startswith("hrfd_debug_")) for n in debug) and len(debug) >= 15
    assert not [n for n in debug if not hasattr(lib, n)]
    assert not [n for n in declared_symbols() if "_debug_" in n], "hrfd.h is the boundary: no debug entry in it"
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = sorted(set(re.findall(r" T (hrfd_[a-z0-9_]+)$", out, flags=re.M)))
    known = set(declared_symbols()) | set(debug)
    assert not [n for n in exported if n not in known], "exported but declared in neither header"


def test_behaviour_changing_hooks_are_inert_without_the_opt_in():
    """a process that did not start with HRFD_DEBUG_HOOKS=1 cannot flip the library onto its test paths: the hooks
    return HRFD_ESTATE before they look at their arguments (so this needs no GPU)"""
    import subprocess, sys
    code = (
        "import ctypes, sys; sys.path.insert(0, %r)\n"
        "from hackrfdiags_amd import _lib\n"
        "L = _lib.load()\n"
        "names = ['hrfd_rx_debug_set_atan', 'hrfd_rx_debug_set_warm', 'hrfd_rx_debug_set_run_len', 'hrfd_rx_debug_set_stream',\n"
        "         'hrfd_rx_debug_expire', 'hrfd_rx_debug_set_fir_flow', 'hrfd_rx_debug_set_gated', 'hrfd_rx_debug_set_stagger',\n"
        "         'hrfd_mod_debug_set_sliced', 'hrfd_mod_debug_set_scan', 'hrfd_mod_debug_set_tail']\n"
        "rcs = [getattr(L, n)(None, 0) for n in names]\n"
        "print(rcs, L.hrfd_last_error().decode())\n" % ROOT)
    env = {k: v for k, v in os.environ.items() if k != "HRFD_DEBUG_HOOKS"}
    off = subprocess.check_output([sys.executable, "-c", code], env=env, text=True)
    assert off.startswith("[-4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4]") and "HRFD_DEBUG_HOOKS" in off
    on = subprocess.check_output([sys.executable, "-c", code], env={**env, "HRFD_DEBUG_HOOKS": "1"}, text=True)
    assert on.startswith("[-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1]")       # HRFD_EINVAL: they looked at the NULL handle


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
    """The forms the kernel uses for the three half-band stages (hrfd_rx_kernels.hip: hb1_bytes on
    bytes with v_lerp_u8; hb2_sum / hb3_sum as 32-bit SWAR with I in bits 0..15 and Q in
    bits 16..31 of one register) equal the reference's Q15 form for every input
    triple, on both fields at once, and no field ever borrows from or carries into
    its neighbour."""
    A = np.arange(-128, 128, dtype=np.int64)
    a, b, c = np.meshgrid(A, A, A, indexing="ij")
    qa, qb, qc = c[::-1, ::-1, ::-1], a[::-1, ::-1, ::-1], b[::-1, ::-1, ::-1]   # a different triple in the Q field

    def pack(i, q):
        return ((i + 128) | ((q + 128) << 16)).astype(np.uint32)

    def direct(h0, x, y, z):
        return (16384 + h0 * (x + z) + 16384 * y) >> 15

    def pk_mad(t, d, k):                                   # v_pk_mad_u16, per 16-bit field
        lo = (t & M(0xFFFF)).astype(np.int64) * d + k
        hi = (t >> M(16)).astype(np.int64) * d + k
        assert lo.max() < 65536 and hi.max() < 65536 and lo.min() >= 0 and hi.min() >= 0
        return (lo | (hi << 16)).astype(np.uint32)

    pa, pb, pc = pack(a, qa), pack(b, qb), pack(c, qc)
    M = np.uint32
    # stage 1 runs on bytes with v_lerp_u8 (per byte (x + y + (r & 1)) >> 1), four bytes per
    # register: hb1_bytes = lerp(lerp(a, c, 0), b, (a ^ c) | (m >> 7)); here on two bytes (I, Q)
    def lerp(x, y, r):
        out = np.zeros_like(x)
        for k in range(4):
            sh = M(8 * k)
            xb, yb, rb = (x >> sh) & M(0xFF), (y >> sh) & M(0xFF), (r >> sh) & M(1)
            out |= (((xb + yb + rb) >> M(1)) & M(0xFF)) << sh
        return out

    ba, bb, bc = (pa & M(0xFF)) | ((pa >> M(8)) & M(0xFF00)), (pb & M(0xFF)) | ((pb >> M(8)) & M(0xFF00)), \
        (pc & M(0xFF)) | ((pc >> M(8)) & M(0xFF00))
    m = lerp(ba, bc, M(0))
    y1 = lerp(m, bb, (ba ^ bc) | (m >> M(7)))
    assert ((y1 & M(0xFF)).astype(np.int64) - 128 == direct(8206, a, b, c)).all()
    assert (((y1 >> M(8)) & M(0xFF)).astype(np.int64) - 128 == direct(8206, qa, qb, qc)).all()
    t = pa + pc
    # stage 2 (its inputs are stage-1 outputs: -128..127 again)
    s2 = t + (pb << M(1)) + ((pk_mad(t, 57, 1792) >> M(13)) & M(0x00070007))
    ac = (s2 >> M(2)) & M(0x00FF00FF)
    assert ((ac & M(0xFFFF)).astype(np.int64) - 128 == direct(8249, a, b, c)).all()
    assert ((ac >> M(16)).astype(np.int64) - 128 == direct(8249, qa, qb, qc)).all()
    assert (((s2 >> M(1)) & M(0x01FE01FE)) == M(2) * ac).all()
    # stage 3: the result is narrowed to int8 (IqDataProcessor.cc:458,489): low byte only
    s3 = t + (pb << M(1)) + ((pk_mad(t, 29, 2816) >> M(10)) & M(0x003F003F)) + M(0x03F803F8)
    y = s3 >> M(2)
    assert ((y & M(0xFF)).astype(np.int64) == ((direct(8424, a, b, c) + 128) & 0xFF)).all()
    assert (((y >> M(16)) & M(0xFF)).astype(np.int64) == ((direct(8424, qa, qb, qc) + 128) & 0xFF)).all()

    # Round 4 forms (hb2_sum / hb3_sum / form_ac with v_pk_lshrrev_b16: a shift per 16-bit field, nothing to mask):
    def pk_shr(v, k):
        return (((v & M(0xFFFF)) >> M(k)) | (((v >> M(16)) >> M(k)) << M(16))).astype(np.uint32)

    def pk_mad_wrap(t, d, k):                              # v_pk_mad_u16 keeps the low 16 bits of every field
        lo = ((t & M(0xFFFF)).astype(np.int64) * d + k) & 0xFFFF
        hi = ((t >> M(16)).astype(np.int64) * d + k) & 0xFFFF
        return (lo | (hi << 16)).astype(np.uint32)

    s2n = t + (pb << M(1)) + pk_shr(pk_mad(t, 57, 1792), 13)
    assert (s2n == s2).all() and (pk_shr(s2n, 2) == ac).all()
    tp = t + M(0x03F803F8)                                 # T' = T - 8 + 4 * 256 per field: no field overflows
    assert ((tp & M(0xFFFF)) < 2048).all() and ((tp >> M(16)) < 2048).all()
    s3n = ((pb << M(1)) + tp) + pk_shr(pk_mad_wrap(tp, 29, 38888), 10)
    assert (s3n == s3).all()
    # the mixer's packed negations act modulo 256 per field, whatever sits above the low byte (s3 >> 2 carries the
    # neighbour's low bits in bits 14..15 of the low field): (0x0100 - field) & 0xFF == (256 - (field & 0xFF)) & 0xFF
    lo = (y & M(0xFFFF)).astype(np.int64)
    assert (((0x100 - lo) & 0xFF) == ((256 - (lo & 0xFF)) & 0xFF)).all()
