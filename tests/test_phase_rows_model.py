"""k_phase_rows' arrangement (and, round 6, k_phase_rows8's: eight steps per lane), modelled on the CPU (numpy, float32 with exactly rounded fused multiply-adds): the Nco phase
recurrence (PhaseAccumulator.cc:157-181) with lane = TIME.  A channel is a row of 16 lanes, lane j holds steps 4j .. 4j+3
of a chunk of 64; in each of sixteen rounds EVERY lane computes x0 = (w3 of its left neighbour) + step0 -- lane 0 keeps the
x0 a rotate gave it from the chunk in front -- and then four wrapped accumulations in its own registers.  The claim the
kernel rests on: after round t lanes 0 .. t are final and stay so (a lane recomputing from a final neighbour gets the same
values), whatever the lanes to their right hold meanwhile; so after sixteen rounds the row holds the 64 accumulators of the
serial recurrence, bit for bit.  (The GPU tests compare the kernel itself with the oracle; this pins the ARGUMENT.)"""
import numpy as np

F = np.float32
K_M, K_CHI, K_CLO = F(float.fromhex("0x1.45f308p-3")), F(float.fromhex("0x1.921fb6p+2")), F(float.fromhex("-0x1.777a5cp-23"))
PI, TWO_PI = 3.14159265358979323846, 6.283185307179586476925286766559


def fma32(a, b, c):
    """float32 fma: the product and the sum are exact in float64 for these operands (a in {-1, 0, 1})"""
    return (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(F)


def wrap_fast(x):
    """hrfd_tx_kernels.hip: k = rint(x M); x = fma(k, -C_HI, x); x = fma(k, -C_LO, x)"""
    k = np.rint(x * K_M).astype(F)
    return fma32(k, -K_CLO, fma32(k, -K_CHI, x))


def wrap_loops(x):
    """the reference's loops (ps_wrap_loops): double compares, double subtraction, rounded to float each turn"""
    x = F(x)
    while float(x) > PI:
        x = F(float(x) - TWO_PI)
    while float(x) < -PI:
        x = F(float(x) + TWO_PI)
    return x


def serial(acc, steps):
    out = np.empty(len(steps), dtype=F)
    acc = F(acc)
    for n, s in enumerate(steps):
        out[n] = acc
        acc = wrap_loops(F(acc + s))
    return out, acc


def rows_chunk(wl, cur, garbage):
    """one chunk of 16 S steps on a row of 16 lanes, S = cur.shape[1] steps per lane (k_phase_rows: 4, k_phase_rows8, round 6:
    8).  wl: the lanes' LAST w registers on entry (lane 15 = the accumulator in front of the chunk, the others whatever the
    last chunk left); cur[lane][r]: the steps; garbage: what the other w registers and x0 hold on entry (anything).  Returns
    the S w registers per lane after sixteen rounds."""
    S = cur.shape[1]
    w = garbage.copy()                                      # [16][S]
    w[:, S - 1] = wl
    x0 = np.roll(w[:, S - 1], 1) + cur[:, 0]                # row_ror:1 add: every lane written, lane 0 from lane 15
    for t in range(16):
        if t:
            shifted = np.roll(w[:, S - 1], 1) + cur[:, 0]   # row_shr:1 add ...
            x0[1:] = shifted[1:]                            # ... lane 0 has no source and keeps its x0
        w[:, 0] = wrap_fast(x0)
        for r in range(1, S):
            w[:, r] = wrap_fast(w[:, r - 1] + cur[:, r])
    return w


import pytest


@pytest.mark.parametrize("S", [4, 8], ids=["k_phase_rows", "k_phase_rows8"])
def test_rows_arrangement_equals_the_serial_recurrence(S):
    rng = np.random.default_rng(7 + S)
    n = 16 * S
    for trial in range(40):
        nchunks = int(rng.integers(1, 6))
        steps = (rng.uniform(-4.85, 4.85, size=n * nchunks)).astype(F)
        if trial % 3 == 0:
            steps[rng.integers(0, steps.size, size=steps.size // 3)] = F(0)
        acc0 = F(rng.uniform(-3.14, 3.14))
        want, want_acc = serial(acc0, steps)
        got = np.empty_like(want)
        wl = np.full(16, acc0, dtype=F)                     # every lane of the row: the phase of cell 0
        garbage = rng.uniform(-50, 50, size=(16, S)).astype(F)
        for i in range(nchunks):
            cur = steps[n * i:n * (i + 1)].reshape(16, S)
            carry = wl[15]
            w = rows_chunk(wl, cur, garbage)
            # lane j's w0 .. w(S-1) are the phases of cells S j + 1 .. S j + S: one cell to the right; cell 0 is the carry
            flat = w.reshape(-1)
            got[n * i] = carry
            got[n * i + 1:n * (i + 1)] = flat[:n - 1]
            wl, garbage = w[:, S - 1].copy(), w.copy()
        assert got.tobytes() == want.tobytes(), trial
        assert wl[15].tobytes() == want_acc.tobytes(), trial


def test_branch_free_wrap_equals_the_loops_where_the_kernel_uses_it():
    """|acc| <= pi and |step| <= 4.85 (the pipeline refuses anything else): spot check of tools/proofs/wrap_rint_fma.c's
    exhaustive claim, around the two boundaries and at random"""
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.uniform(-8, 8, 20000).astype(F),
                         np.nextafter(F(PI), F(0)) + np.arange(-200, 200).astype(F) * np.spacing(F(PI)),
                         -(np.nextafter(F(PI), F(0)) + np.arange(-200, 200).astype(F) * np.spacing(F(PI)))]).astype(F)
    fast = wrap_fast(xs)
    slow = np.array([wrap_loops(x) for x in xs], dtype=F)
    assert ((fast == slow) | ((fast == 0) & (slow == 0))).all()
