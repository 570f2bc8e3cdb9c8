"""Developer check run on the GPU box: parity of the HIP path vs the CPU oracle
on a few cases + a first timing.  Not part of the test suite."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hackrfdiags_amd import api, synth  # noqa: E402
from tests.reflib import Oracle, WBFM, NONE  # noqa: E402

BLK = synth.BLOCK_BYTES
orc = Oracle()


def cmp(name, got, want):
    got = np.asarray(got); want = np.asarray(want)
    if got.shape != want.shape:
        print(f"  {name}: SHAPE {got.shape} vs {want.shape}")
        return False
    bad = np.nonzero(got != want)[0] if got.ndim == 1 else np.argwhere(got != want)
    if len(bad) == 0:
        print(f"  {name}: exact ({got.size} values)")
        return True
    d = np.abs(got.astype(np.int64) - want.astype(np.int64))
    print(f"  {name}: {len(bad)} mismatches of {got.size}, max |diff| {d.max()}, first at {bad[:5].tolist()}")
    return False


def single(kind, mode, nblk=3, warm=None):
    print(f"[single channel] kind={kind} mode={mode} warm={warm}")
    x = synth.make_input(kind, 3, nblk)
    g = api.SingleChannelRx(); g.set_mode(mode)
    if warm is not None:
        g.rx.debug_set_warm(warm)
    o = orc.rx(); o.set_mode(mode)
    ok = True
    for b in range(nblk):
        pg, mg, ag, ig = g.process(x[b * BLK:(b + 1) * BLK])
        po, mo, ao, io = o.process(x[b * BLK:(b + 1) * BLK])
        ok &= cmp(f"blk{b} iq256", ig, io)
        ok &= (mg == mo) or print("  mag", mg, mo)
        ok &= cmp(f"blk{b} pcm", pg, po)
    print("  counters", g.rx.debug_counters())
    return ok


def batch(kind, C, B, warm=None):
    print(f"[batch] kind={kind} C={C} B={B} warm={warm}")
    xs = np.stack([synth.make_input(kind, 10 + c, B) for c in range(C)]).reshape(C, B, BLK)
    rx = api.Rx(C); rx.set_mode(WBFM)
    if warm is not None:
        rx.debug_set_warm(warm)
    t0 = time.time()
    pcm, n_pcm, mag, allowed, iq256 = rx.process_block(xs, B, want_iq256=True)
    t1 = time.time()
    print(f"  host call {1e3 * (t1 - t0):.1f} ms; counters {rx.debug_counters()}")
    ok = True
    for c in range(C):
        o = orc.rx(); o.set_mode(WBFM)
        want = [o.process(xs[c, b]) for b in range(B)]
        ok &= cmp(f"ch{c} iq256", iq256[c].reshape(-1), np.concatenate([w[3] for w in want]))
        ok &= cmp(f"ch{c} pcm", pcm[c].reshape(-1), np.concatenate([w[0] for w in want]))
        ok &= cmp(f"ch{c} mag", mag[c], np.array([w[1] for w in want], dtype=np.uint32))
    # second call continues the streams
    xs2 = np.stack([synth.make_input(kind, 10 + c, 2 * B)[B * BLK:] for c in range(C)]).reshape(C, B, BLK)
    xs_all = np.stack([synth.make_input(kind, 10 + c, 2 * B) for c in range(C)]).reshape(C, 2 * B, BLK)
    pcm2 = rx.process_block(xs2, B)[0]
    for c in range(min(C, 2)):
        o = orc.rx(); o.set_mode(WBFM)
        want = np.concatenate([o.process(xs_all[c, b])[0] for b in range(2 * B)])
        ok &= cmp(f"ch{c} pcm (2nd call)", pcm2[c].reshape(-1), want[B * 512:])
    return ok


def timing(C, B, reps=5):
    import torch
    print(f"[timing] C={C} B={B}")
    dev = torch.device("cuda:0")
    x = torch.randint(-128, 128, (C, B, BLK), dtype=torch.int8, device=dev)
    pcm = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
    rx = api.Rx(C); rx.set_mode(WBFM)
    torch.cuda.synchronize()
    for _ in range(2):
        rx.process_device(x.data_ptr(), B * BLK, BLK, B, pcm.data_ptr())
        v = rx.sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        rx.process_device(x.data_ptr(), B * BLK, BLK, B, pcm.data_ptr())
        v = rx.sync()
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    nbytes = C * B * BLK
    print(f"  violations {v}; best {1e3 * t:.3f} ms -> {nbytes / t / 1e9:.1f} GB/s, {nbytes / 2 / t / 1e6:.0f} MS/s; counters {rx.debug_counters()}")


if __name__ == "__main__":
    print("devices:", api.device_count())
    ok = True
    ok &= single("lcg", NONE, 2)
    ok &= single("lcg", WBFM)
    ok &= single("fmtone", WBFM)
    ok &= single("dc_pos", WBFM, 2)
    ok &= batch("lcg", 3, 4)
    ok &= batch("fmtone", 9, 3)
    print("---- forced speculation misses (short warm-up): repairs / replays must keep it exact")
    ok &= single("fmtone", WBFM, 3, warm=320)
    ok &= single("lcg", WBFM, 3, warm=64)
    ok &= batch("lcg", 3, 4, warm=384)
    ok &= batch("fmtone", 5, 3, warm=128)
    timing(64, 16)
    timing(256, 16)
    print("ALL OK" if ok else "SOME FAILED")
