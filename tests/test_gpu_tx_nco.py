"""GPU parity of the transmit mirror (SSB modulator, interpolateSignal cascade) and
the Nco against golden vectors and the CPU oracle."""
import numpy as np
import pytest

from hackrfdiags_amd import api, synth
from tests import goldencheck as G

pytestmark = pytest.mark.gpu
ARR, MAN = G.load()


@pytest.fixture(scope="module")
def engine():
    if api.device_count() < 1:
        pytest.fail("no GPU visible: the HIP path cannot run (and there is no CPU fallback)")
    return api.Engine()


@pytest.mark.parametrize("case", MAN["tx"], ids=lambda c: c["key"])
def test_golden_ssb_modulator(engine, case):
    G.check_tx_case(engine, ARR, case)


def test_golden_interpolate_signal(engine):
    G.check_interp(engine, ARR, MAN)


@pytest.mark.parametrize("lsb", [True, False])
def test_ssb_modulator_batched_and_ragged(oracle, lsb):
    """many channels, several calls of different lengths (incl. not a multiple of the
    32-sample tile), sideband switch and reset between calls"""
    C = 5
    pcm = np.stack([synth.lcg_pcm(7 + c, 3000) for c in range(C)])
    m = api.Mod(api.MOD_SSB, C)
    m.set_sideband(lsb)
    os_ = [oracle.ssbmod(lsb) for _ in range(C)]
    off = 0
    for k, n in enumerate([512, 512, 100, 33, 1, 700, 512]):
        if k == 3:
            m.set_sideband(not lsb)
            for o in os_:
                o.set_sideband(not lsb)
        if k == 5:
            m.reset()
            for o in os_:
                o.reset()
        got = m.process(pcm[:, off:off + n])
        for c in range(C):
            want = np.concatenate([os_[c].process(pcm[c, off + s:off + min(s + 512, n)]) for s in range(0, n, 512)])
            assert (got[c] == want).all(), (k, n, c)
        off += n


def test_interp_batched(oracle):
    C = 3
    iq = np.stack([synth.lcg_pcm(90 + c, 2 * 900) for c in range(C)])
    m = api.Mod(api.MOD_INTERP, C)
    os_ = [oracle.interp() for _ in range(C)]
    for lo, hi in [(0, 256), (256, 300), (300, 900)]:
        got = m.process(iq[:, 2 * lo:2 * hi])
        for c in range(C):
            assert (got[c] == os_[c].process(iq[c, 2 * lo:2 * hi])).all()


def test_modulator_full_scale_and_device_entry(oracle):
    import torch
    C, n = 4, 1024
    pcm = np.stack([np.full(n, v, dtype=np.int16) for v in (32767, -32768, 0, 12345)])
    pcm[3, ::2] = -32768
    dev = torch.device("cuda:0")
    d_in = torch.from_numpy(pcm).to(dev)
    d_out = torch.zeros((C, 512 * n), dtype=torch.int8, device=dev)
    torch.cuda.synchronize()                             # torch filled these on its own stream
    m = api.Mod(api.MOD_SSB, C)
    m.process_device(d_in.data_ptr(), n, d_out.data_ptr())
    m.sync()
    got = d_out.cpu().numpy()
    for c in range(C):
        o = oracle.ssbmod(True)
        want = np.concatenate([o.process(pcm[c, s:s + 512]) for s in range(0, n, 512)])
        assert (got[c] == want).all()


def test_nco_fast_and_run_are_bit_exact(oracle):
    bits = lambda a: np.ascontiguousarray(a).view(np.int32).astype(np.int64)
    for fs, f in [(8000.0, 1000.0), (256000.0, 75000.0), (256000.0, -12345.6)]:
        g, o = api.Nco(fs, f, 1), oracle.nco(fs, f)
        ia, qa = g.run(3000, fast=True)
        ib, qb = o.run(3000, True)
        assert (bits(ia) == bits(ib)).all() and (bits(qa) == bits(qb)).all()     # runFast: table lookup, exact
        g.set_frequency(f / 3); o.set_frequency(f / 3)
        ia, qa = g.run(2000, fast=False)
        ib, qb = o.run(2000, False)
        # Nco::run calls libm sinf/cosf (Nco.cc:186-199): glibc's algorithm restated on the device (round 5: glibc_sinf /
        # glibc_cosf, checked against the host's libm on every float by tools/proofs/sincosf_glibc.c) -- no tolerance
        assert (bits(ia) == bits(ib)).all() and (bits(qa) == bits(qb)).all()
        g.reset(); o.reset()
        ia, _ = g.run(10, True); ib, _ = o.run(10, True)
        assert (bits(ia) == bits(ib)).all()


def test_nco_many_channels(oracle):
    C = 6
    g = api.Nco(256000.0, 1000.0, C)
    for c in range(C):
        g.set_frequency(1000.0 * (c + 1), channel=c)
    i, q = g.run(500, fast=True)
    for c in range(C):
        o = oracle.nco(256000.0, 1000.0)
        o.set_frequency(1000.0 * (c + 1))
        ib, qb = o.run(500, True)
        assert (i[c].view(np.uint32) == ib.view(np.uint32)).all()
        assert (q[c].view(np.uint32) == qb.view(np.uint32)).all()


# ---------------------------------------------------------------- AM / FM modulators
def _mod_case(oracle, kind, api_kind, tol, tail=None):
    """many channels, calls of ragged lengths, a per-channel parameter change and a reset between
    calls; returns the fraction of output bytes that differ from the oracle (all within tol)."""
    C = 4
    pcm = np.stack([synth.lcg_pcm(17 + c, 2400) if c % 2 == 0 else
                    np.round(25000 * np.sin(2 * np.pi * (300 + 170 * c) * np.arange(2400) / 8000)).astype(np.int16)
                    for c in range(C)])
    pcm[3, 100:140] = 32767
    pcm[3, 140:180] = -32768
    m = api.Mod(api_kind, C)
    if tail is not None:
        m.debug_set_tail(tail)
    os_ = [getattr(oracle, kind)() for _ in range(C)]
    off, diff, total = 0, 0, 0
    for k, n in enumerate([512, 512, 100, 33, 1, 700, 512]):
        if k == 2:
            v = {"ammod": 0.4, "fmmod": 1500.0, "wbfmmod": 30000.0}[kind]
            m.set_param(v, channel=1)
            os_[1].set_param(v)
        if k == 5:
            m.reset()
            for o in os_:
                o.reset()
        got = m.process(pcm[:, off:off + n])
        for c in range(C):
            want = os_[c].process(pcm[c, off:off + n])
            d = np.abs(got[c].astype(np.int16) - want.astype(np.int16))
            # int8 wrap-around is part of the contract: compare modulo 256
            d = np.minimum(d, 256 - d)
            assert d.max() <= tol, (kind, k, n, c, int(d.max()))
            diff += int((d != 0).sum()); total += d.size
        off += n
    return diff / total


def test_am_modulator_bit_exact(oracle):
    assert _mod_case(oracle, "ammod", api.MOD_AM, 0) == 0.0


@pytest.mark.parametrize("tail", [1, 0], ids=["k_wb_tail", "rails_then_cascade"])
def test_wbfm_modulator_bit_exact(oracle, tail):
    """WbFmModulator: integer cascades around a table-lookup Nco whose phase recurrence is
    reproduced operation for operation -- no tolerance.  Both forms of the last pass: the lookup fused into the x8 cascade
    (k_wb_tail, round 6, the default) and rounds 2-5's two kernels."""
    assert _mod_case(oracle, "wbfmmod", api.MOD_WBFM, 0, tail) == 0.0


@pytest.mark.parametrize("n", [16 * 512 + 77, 2061, 1601, 1537, 4600, 4609], ids=lambda n: "n%d" % n)
def test_wbfm_modulator_time_slices(oracle, n):
    """a long call runs its passes in time slices on two streams of the handle's own -- the phase recurrence of slice
    t + 1 beside the table lookup and the x8 cascade of slice t: bit-exact against the oracle, equal to the unsliced call
    (hook), and the state a second call continues from.  Lengths: 16 blocks and a ragged rest (nine slices), the
    shortest call that is sliced at all (25 tiles: 1537 samples) and its neighbours, and both sides of the length from
    which the slices halve four times at the end instead of twice (72 tiles)."""
    C = 5
    pcm = np.stack([synth.lcg_pcm(140 + c, 2 * n) for c in range(C)])
    a, b, d = api.Mod(api.MOD_WBFM, C), api.Mod(api.MOD_WBFM, C), api.Mod(api.MOD_WBFM, C)
    a.debug_set_sliced(2)                                   # slices whether or not the recurrence's stream got CUs of its own
    b.debug_set_sliced(0)
    d.debug_set_sliced(2)
    d.debug_set_tail(0)                                     # rounds 2-5's last pass: k_wb_rails in place, then k_mod<WB_TAIL>
    os_ = [oracle.wbfmmod() for _ in range(C)]
    for call in range(2):
        x = pcm[:, call * n:(call + 1) * n]
        ga, gb, gd = a.process(x), b.process(x), d.process(x)
        assert (ga == gb).all(), call
        assert (ga == gd).all(), call
        for c in range(C):
            assert (ga[c] == os_[c].process(x[c])).all(), (call, c)


def test_wbfm_modulator_absurd_deviation_takes_the_loops(oracle):
    """setFrequencyDeviation tests the CURRENT value against its limit (WbFmModulator.cc:313), so one absurd deviation
    gets through: Nco steps of tens of radians, several turns of the wrap loops per sample.  k_phase_scan's branch-free
    wrap is proven for |acc + step| <= 8 only; chunks with larger steps must come out of the loops, bit-exact, and the
    chunks of the other channels in the same workgroup with them.  17 channels: two workgroups, the second with one."""
    C = 17
    n = 700
    pcm = np.stack([synth.lcg_pcm(90 + c, n) for c in range(C)])
    m = api.Mod(api.MOD_WBFM, C)
    os_ = [oracle.wbfmmod() for _ in range(C)]
    for c, dev in ((2, 2.0e6), (16, 9.0e5)):
        m.set_param(dev, channel=c)
        os_[c].set_param(dev)
    off = 0
    for k in (300, 1, 399):
        got = m.process(pcm[:, off:off + k])
        for c in range(C):
            want = os_[c].process(pcm[c, off:off + k])
            assert (got[c] == want).all(), (k, c)
        off += k


def test_fm_modulator_bit_exact(oracle):
    """FmModulator's Nco calls libm cosf/sinf (Nco.cc:186-199, FmModulator.cc:600-603).  Rounds 1-4 evaluated cos / sin in
    double and rounded (+-1 LSB of the int8 IQ, BASELINE.json's allowance for the trig paths); round 5 restates glibc's
    sinf / cosf on the device: every byte is the oracle's."""
    assert _mod_case(oracle, "fmmod", api.MOD_FM, 0) == 0.0


ARR_MOD, MAN_MOD = G.load_mod()


@pytest.mark.parametrize("n", [8192, 4096, 4096 + 64 + 3, 4032, 8192 + 512 + 7])
def test_fm_modulator_time_slices(oracle, n):
    """Round 4: a call of 64 tiles or more (4096 PCM samples) runs in three time slices -- the phase recurrence and the
    cos / sin pass of every slice on the handle's own stream ahead of the cascade launch of the slice in front.  The
    sliced call must equal the unsliced one (hook) BYTE FOR BYTE (same kernels, same arithmetic, only the order of
    launches differs), the oracle's bytes, and leave the state a second call continues from.
    Lengths: 16 blocks, the shortest call that is sliced and its lower neighbour (unsliced), ragged tails (the last
    slice's recurrence then runs on the plain kernel)."""
    C = 5
    pcm = np.stack([synth.lcg_pcm(170 + c, 2 * n) for c in range(C)])
    a, b = api.Mod(api.MOD_FM, C), api.Mod(api.MOD_FM, C)
    b.debug_set_sliced(0)
    for m in (a, b):
        m.set_param(2500.0, channel=1)
    os_ = [oracle.fmmod() for _ in range(C)]
    os_[1].set_param(2500.0)
    for call in range(2):
        x = pcm[:, call * n:(call + 1) * n]
        ga, gb = a.process(x), b.process(x)
        assert (ga == gb).all(), call
        for c in range(C):
            want = os_[c].process(x[c])
            assert (ga[c] == want).all(), (call, c)


@pytest.mark.parametrize("kind", ["fm", "wbfm"])
@pytest.mark.parametrize("C", [1, 3, 4, 5, 17, 67])
def test_phase_recurrence_kernels_agree(oracle, kind, C):
    """Round 4: the Nco phase recurrence runs on k_phase_rows (lane = time: a channel is a row of 16 lanes, four channels per
    wave, the accumulator handed along the row by DPP) and no longer on k_phase_scan (lane = channel, cells through LDS).
    Both must give the same bytes (hook: debug_set_scan(1) is the old kernel), call after call, for banks that fill a wave,
    leave rows of the last wave empty (they repeat the bank's last channel) or take several workgroups, with one channel
    at an absurd deviation (WBFM: its chunks are refused by the pipeline and go through the reference's loops, the chunks
    around them through the pipeline again) -- and the oracle's, exactly."""
    mk = api.MOD_FM if kind == "fm" else api.MOD_WBFM
    n = 1024 + 512
    pcm = np.stack([synth.lcg_pcm(300 + c, 2 * n) for c in range(C)])
    a, b, d = api.Mod(mk, C), api.Mod(mk, C), api.Mod(mk, C)
    b.debug_set_scan(1)
    d.debug_set_scan(2)                                     # round 6: a is k_phase_rows8 (eight steps per lane) for WBFM, d the four-step kernel
    os_ = [getattr(oracle, kind + "mod")() for _ in range(C)]
    if kind == "wbfm" and C >= 3:
        for m in (a, b, d):
            m.set_param(1.5e6, channel=C - 2)
        os_[C - 2].set_param(1.5e6)
    for call in range(2):
        x = pcm[:, call * n:(call + 1) * n]
        ga, gb, gd = np.atleast_2d(a.process(x)), np.atleast_2d(b.process(x)), np.atleast_2d(d.process(x))
        assert (ga == gb).all(), call
        assert (ga == gd).all(), call
        for c in range(C):
            want = os_[c].process(x[c])
            assert (ga[c] == want).all(), (call, c)


@pytest.mark.parametrize("C", [4100, 8200])
def test_wbfm_modulator_big_banks(oracle, C):
    """the phase recurrence's two shapes at bank sizes that matter to them: 4100 channels (k_phase_rows with more than one
    wave on some SIMDs: 257 workgroups of four waves) and 8200 (beyond 8192: round 2's k_phase_scan<64>, 64 channels per
    recurrence wave).  One block per channel, every 97th channel and the bank's last against the oracle, bit for bit."""
    n = 512
    rng = np.random.default_rng(C)
    pcm = rng.integers(-20000, 20000, size=(C, n), dtype=np.int16)
    m = api.Mod(api.MOD_WBFM, C)
    got = m.process(pcm)
    for c in list(range(0, C, 97)) + [C - 2, C - 1]:
        assert (got[c] == oracle.wbfmmod().process(pcm[c])).all(), c


@pytest.mark.parametrize("case", MAN_MOD["am"], ids=lambda c: c["key"])
def test_golden_am_modulator(engine, case):
    G.check_am_mod(engine, ARR_MOD, case)


@pytest.mark.parametrize("case", MAN_MOD["fm"], ids=lambda c: c["key"])
def test_golden_fm_modulator(engine, case):
    G.check_fm_mod(engine, ARR_MOD, case, tol=0)     # round 5: the reference's bytes and digest (glibc cosf / sinf restated on the device)


@pytest.mark.parametrize("case", MAN_MOD["wbfm"], ids=lambda c: c["key"])
def test_golden_wbfm_modulator(engine, case):
    G.check_wbfm_mod(engine, ARR_MOD, case)


@pytest.mark.parametrize("seed", list(range(1, 1 + int(__import__("os").environ.get("HRFD_WALK_SEEDS", "6")))))
def test_random_walk_of_modulator_calls(oracle, seed):
    """The transmit mirror of the receive side's random walk: banks of 1..40 modulators of one kind, calls of random length
    (1 sample .. beyond the 64-tile mark from which the FM and WBFM modulators run in time slices), parameter changes,
    sideband switches and resets between calls; every channel against its own oracle object.  Bit-exact, every kind."""
    rng = np.random.default_rng(5000 + seed)
    kind = ["ssb", "am", "fm", "wbfm"][seed % 4]
    C = int(rng.integers(1, 41))
    amod = {"ssb": api.MOD_SSB, "am": api.MOD_AM, "fm": api.MOD_FM, "wbfm": api.MOD_WBFM}[kind]
    m = api.Mod(amod, C)
    if kind == "ssb":
        lsb = [True] * C
        os_ = [oracle.ssbmod(True) for _ in range(C)]
    else:
        os_ = [getattr(oracle, kind + "mod")() for _ in range(C)]
    total = 0
    for call in range(5):
        n = int(rng.choice([1, 33, 512, 700, 1537, 4096, 4163, 6000, 8192]))
        if kind == "wbfm":
            n = min(n, 4163)                               # (the oracle's x32 stage is slow: keep the walk short)
        pcm = np.stack([synth.lcg_pcm(int(rng.integers(0, 10000)), n) if rng.random() < 0.8 else np.full(n, int(rng.integers(-32768, 32768)), dtype=np.int16)
                        for _ in range(C)])
        for c in range(C):
            r = rng.random()
            if r < 0.15:
                m.reset(channel=c); os_[c].reset()
            elif r < 0.35:
                if kind == "ssb":
                    lsb[c] = not lsb[c]
                    m.set_sideband(lsb[c], channel=c); os_[c].set_sideband(lsb[c])
                else:
                    v = {"am": float(rng.choice([0.0, 0.3, 0.8, 1.0])), "fm": float(rng.choice([500.0, 1200.0, 3500.0])),
                         "wbfm": float(rng.choice([10000.0, 30000.0, 70000.0, 112000.0]))}[kind]
                    m.set_param(v, channel=c); os_[c].set_param(v)
        got = np.atleast_2d(m.process(pcm))                 # (one channel: api.Mod.process returns its row)
        for c in range(C):
            if kind == "ssb":
                want = np.concatenate([os_[c].process(pcm[c, s:min(s + 512, n)]) for s in range(0, n, 512)])
            else:
                want = os_[c].process(pcm[c])
            assert (got[c] == want).all(), (seed, kind, call, c, n)
        total += n
