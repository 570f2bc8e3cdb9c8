"""CPU models of the two things round 5's re-split of k_rx_wbfm_flow rests on (hackrfdiags_amd/csrc/hrfd_rx_flow.hip:
service_waves_wbfm2, hrfd_rx_kernels.hip: theta_quad), in numpy float32 -- no GPU.

1. The first-quadrant atan2 table the library builds (and proves) at hrfd_rx_create, fetched through the host-only
   entry hrfd_debug_atan2_quadrant: theta_quad's arithmetic on it, restated here, reproduces ALL 65536 entries of the
   reference's table (WbFmDemodulator.cc:137-148: hrfd_atan2_table) bit for bit.
2. The warm-up scheme: round 4's kernel had lane l run tiles l - 2 and l - 1 out of the ring itself before its own tile;
   the re-split kernel keeps a tile's v in registers and makes two passes over the lane's OWN tile, each started from
   the value the LEFT lane's previous pass ended with.  Both give every lane the same start value, bit for bit -- so the
   verification, the repairs and the PCM are the same -- including the stream's first lanes (exact start from the
   carried y) and the hand-over between generations."""
import ctypes as C

import numpy as np

from hackrfdiags_amd import _lib, api

F = np.float32
A1 = F(-0.9492274)


def test_first_quadrant_table_reproduces_the_reference_table():
    L = _lib.load()
    tq = np.zeros(16644, dtype=np.uint32)
    ok = C.c_int(0)
    assert L.hrfd_debug_atan2_quadrant(tq.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(ok)) == 0
    assert ok.value == 1, "the first-quadrant table did not verify against the reference table on this libm"
    want = api.atan2_table().view(np.uint32)               # [q_idx][i_idx]
    qi, ii = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    q, i = qi - 128, ii - 128
    w = tq[np.abs(q) * 129 + np.abs(i)]
    t = w & np.uint32(0x3FFFFFFF)
    fix = (w.astype(np.int64) >> 30)
    fix = np.where(fix >= 2, fix - 4, fix)                 # the two top bits as a signed field
    with np.errstate(over="ignore"):
        pv = ((F(3.14159274) - t.view(np.float32)).astype(np.float32).view(np.uint32).astype(np.int64) + fix).astype(np.uint32)
    mag = np.where(i < 0, pv, t)
    got = np.where(q < 0, mag | np.uint32(0x80000000), mag)
    assert (got == want).all(), int((got != want).sum())
    assert set(np.unique(fix)) <= {-2, -1, 0, 1}


def _run_tile(v, y):
    for x in v:
        r = F(A1 * y)
        y = F(x - r)
    return y


def test_own_tile_passes_equal_the_ring_warm_up():
    rng = np.random.default_rng(5)
    n_tiles, T, wt, M = 40, 64, 2, 5
    v = (rng.standard_normal((n_tiles, T)) * 900).astype(np.float32)
    y_in = F(123.456)
    c = F(-A1)
    ct = F(float(c) ** T)
    # geometric partial sums per tile (approximate on purpose; both schemes use the same ones)
    P = np.zeros(n_tiles, dtype=np.float32)
    for t in range(n_tiles):
        acc = F(0)
        for x in v[t]:
            acc = F(acc * c + x)
        P[t] = acc
    P[0] = F(P[0] + F(float(c) ** T) * y_in)

    def seed(ws):
        acc = F(0)
        for m in range(M, 0, -1):
            idx = ws - m
            acc = F(acc * ct + (P[idx] if idx >= 0 else F(0)))
        return acc

    # round 4: lane t starts from the seed at the end of tile t - wt - 1 and runs tiles t - wt .. t - 1 itself
    old = np.zeros(n_tiles, dtype=np.float32)
    for t in range(n_tiles):
        ws = t - wt
        y = y_in if ws <= 0 else seed(ws)
        for k in range(wt, 0, -1):
            if t - k >= 0:
                y = _run_tile(v[t - k], y)
        old[t] = y
    # round 5: wt passes over the lane's own tile, pass k from the left lane's pass k - 1 (lane 0 of the stream: the carried y)
    start = np.array([y_in if t == 0 else seed(t) for t in range(n_tiles)], dtype=np.float32)
    for _ in range(wt):
        ends = np.array([_run_tile(v[t], start[t]) for t in range(n_tiles)], dtype=np.float32)
        start = np.concatenate([[y_in], ends[:-1]]).astype(np.float32)
        start[0] = y_in
    assert (start.view(np.uint32) == old.view(np.uint32)).all()
