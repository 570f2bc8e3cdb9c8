"""first-launch counters of one shape through the flow kernel (device path, no replay)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hackrfdiags_amd import api, synth
bb, C, B, run_len, kern = [int(a) for a in sys.argv[1:6]]
BLK = synth.BLOCK_BYTES
raw = np.concatenate([synth.make_input("fmtone" if c % 2 else "lcg", 120 + c, (B * bb + BLK - 1) // BLK)[: B * bb]
                      for c in range(C)]).reshape(C, B, bb)
x = torch.from_numpy(raw).cuda()
pcm = torch.zeros((C, B, bb // 512), dtype=torch.int16, device="cuda")
mag = torch.zeros((C, B), dtype=torch.int32, device="cuda")
rx = api.Rx(C); rx.set_mode(api.WBFM); rx.debug_set_stream(kern); rx.debug_set_run_len(run_len)
grid = 8 * ((C + 7) // 8) * B
rx.debug_stamps(grid)
rx.process_device(x.data_ptr(), B * bb, bb, B, pcm.data_ptr(), d_magnitude=mag.data_ptr())
v = rx.sync()
print("sync ->", v, "counters [repairs, gate, spec, commit, ...]", rx.debug_counters())
print("magnitudes", mag.cpu().numpy().tolist())
import ctypes
pub = np.zeros(C * B, dtype=np.float32); spec = np.zeros(C * B, dtype=np.float32)
rx.L.hrfd_rx_debug_chk.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
rx.L.hrfd_rx_debug_chk(rx.h, pub.ctypes.data, spec.ctypes.data, C * B)
pub = pub.reshape(C, B).view(np.uint32); spec = spec.reshape(C, B).view(np.uint32)
for c in range(C):
    print("  pub ", " ".join(f"{v:08x}" for v in pub[c]))
    print("  spec", " ".join(f"{v:08x}" for v in spec[c]))
from tests.reflib import Oracle, WBFM
got = pcm.cpu().numpy()
for c in range(C):
    o = Oracle().rx(); o.set_mode(WBFM)
    for b in range(B):
        p, m, _, _ = o.process(raw[c, b])
        d = np.nonzero(got[c, b, :len(p)] != p)[0]
        if len(d):
            print(f"  c {c} b {b}: {len(d)} PCM samples differ, first {d[:6].tolist()} last {d[-3:].tolist()}")
st = rx.debug_stamps(grid, read=True)
for w in range(grid):
    if st[w, 40] or st[w, 41]:
        print(f"  wg {w}: spec tile {int(st[w,40])>>32} y {int(st[w,40])&0xffffffff:08x} (blk {int(st[w,42])>>32}) | pub tile {int(st[w,41])>>32} y {int(st[w,41])&0xffffffff:08x} (blk {int(st[w,43])>>32})")
for w in np.nonzero(st[:, 7])[0]:
    print("   finalize at", int(st[w, 44]), "by wave/unit", int(st[w, 47]) >> 16, int(st[w, 47]) & 0xffff, "| wave 4 wait from", int(st[w, 45]), "to", int(st[w, 46]))
    print("   dbg", [int(v) for v in st[w, 1:6]], f"workgroup {w}: wait {int(st[w, 7]) >> 32} expired in wave {(int(st[w, 7]) & 0xffffffff) - 1}")
