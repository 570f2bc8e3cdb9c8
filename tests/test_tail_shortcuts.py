"""The x8 tail's instruction diet of round 6 (k_mod / k_wb_tail: tail_eight, csrc/hrfd_tx_kernels.hip), pinned on the CPU:
the three rewritings are IDENTITIES on the byte that is kept -- byte 2 of a word whose bits 16..23 are the low byte of
the stage-8 output (SsbModulator.cc:607-610: the (int8_t) narrowing) -- over everything their inputs can be.

  (1) b = N >> 15 is a phase-0 output of stage 7, N = 16384 + H2 (x + y); its phase-1 successor in stage 8 is
      (b + 1) >> 1: byte 2 of (b << 15) + (1 << 15).  Claim: that is byte 2 of N + (1 << 15).     All N a sum can make.
  (2) b = (a + 1) >> 1 is a phase-1 output; c = (a + 3) >> 1 = b + 1.  Claims: byte 2 of c << 15 is byte 2 of
      (b << 15) + (1 << 15), and ((1 << 15) - 2 H1) + 2 H1 (c + y) = (1 << 15) + 2 H1 (b + y) in int32.   All int16 a, y.
  (3) the whole tail, round 5's form against round 6's, on extreme and random (x5[j], x5[j-1]) pairs, int32 wrap-around.
"""
import re
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def taps():
    txt = open(os.path.join(ROOT, "hackrfdiags_amd", "csrc", "hrfd_tables.h")).read()
    out = {}
    for name in ("Q_INTERP_HB1", "Q_INTERP_HB2", "Q_INTERP_HB3"):
        m = re.search(name + r"\[\d*\]\s*=\s*\{([^}]*)\}", txt)
        out[name] = int(m.group(1).split(",")[0])
    return out["Q_INTERP_HB1"], out["Q_INTERP_HB2"], out["Q_INTERP_HB3"]


H1, H2, H3 = taps()
I32 = np.int32


def byte2(z):
    return (z.astype(np.int64) >> 16) & 0xff


def test_taps_are_the_ones_the_claims_were_made_for():
    assert (H3, H2, H1) == (8424, 8249, 8206)             # stages 6, 7, 8 (SsbModulator.cc interpolators 6, 7, 8)


def test_phase1_of_a_phase0_output_straight_from_its_numerator():
    s = np.arange(-65536, 65535, dtype=np.int64)            # x + y, both int16
    n = (16384 + H2 * s).astype(I32)
    b = n >> 15
    old = ((b << 15) + (1 << 15)).astype(I32)
    new = (n + (1 << 15)).astype(I32)
    assert (byte2(old) == byte2(new)).all()
    assert (byte2(new) == (((b + 1) >> 1) & 0xff)).all()    # ... and it IS the low byte of the reference's output


def test_phase1_inputs_carried_plus_one():
    a = np.arange(-32768, 32768, dtype=np.int64)
    b, c = (a + 1) >> 1, (a + 3) >> 1
    assert (c == b + 1).all()
    assert (byte2((c << 15).astype(I32)) == byte2(((b << 15) + (1 << 15)).astype(I32))).all()
    y = np.arange(-32768, 32768, 97, dtype=np.int64)
    bb, yy = np.meshgrid(b, y, indexing="ij")
    old = ((1 << 15) + 2 * H1 * (bb + yy)).astype(I32)
    new = (((1 << 15) - 2 * H1) + 2 * H1 * ((bb + 1) + yy)).astype(I32)
    assert (old == new).all()


def tail_round5(xa, xb):
    """one rail: x5[j], x5[j-1] -> the eight z words (byte 2 = the output), as rounds 2-5 computed them"""
    def hb4(h, xn, xm1):
        return ((1 << 14) + h * (xn + xm1)) >> 15, (xn + 1) >> 1
    def hb4z(xn, xm1):
        return (1 << 15) + 2 * H1 * (xn + xm1), (xn << 15) + (1 << 15)
    p1 = (xb + 1) >> 1
    a0, a1 = hb4(H3, xa, xb)
    q1 = (p1 + 1) >> 1
    b0, b1 = hb4(H2, a0, p1)
    b2, b3 = hb4(H2, a1, a0)
    z = [*hb4z(b0, q1), *hb4z(b1, b0), *hb4z(b2, b1), *hb4z(b3, b2)]
    return [v.astype(I32) for v in z]


def tail_round6(xa, xb):
    p1 = (xb + 1) >> 1
    a0, a1 = ((1 << 14) + H3 * (xa + xb)) >> 15, (xa + 1) >> 1
    q1 = (p1 + 1) >> 1
    n0, n2 = (1 << 14) + H2 * (a0 + p1), (1 << 14) + H2 * (a1 + a0)
    b0, b2 = n0 >> 15, n2 >> 15
    c1, c3 = (a0 + 3) >> 1, (a1 + 3) >> 1
    k = (1 << 15) - 2 * H1
    z = [(1 << 15) + 2 * H1 * (b0 + q1), n0 + (1 << 15), k + 2 * H1 * (c1 + b0), c1 << 15,
         k + 2 * H1 * (b2 + c1), n2 + (1 << 15), k + 2 * H1 * (c3 + b2), c3 << 15]
    return [v.astype(I32) for v in z]


def test_whole_tail_round5_form_equals_round6_form():
    rng = np.random.default_rng(11)
    edge = np.array([-32768, -32767, -16385, -16384, -1025, -1024, -513, -512, -2, -1, 0, 1, 2, 511, 512, 1023, 1024, 16383, 16384, 32766, 32767], dtype=np.int64)
    ea, eb = np.meshgrid(edge, edge, indexing="ij")
    xa = np.concatenate([ea.ravel(), rng.integers(-32768, 32768, 2_000_000), rng.integers(-1200, 1200, 1_000_000)])
    xb = np.concatenate([eb.ravel(), rng.integers(-32768, 32768, 2_000_000), rng.integers(-1200, 1200, 1_000_000)])
    for zo, zn in zip(tail_round5(xa, xb), tail_round6(xa, xb)):
        assert (byte2(zo) == byte2(zn)).all()
