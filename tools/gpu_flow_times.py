"""k_rx_wbfm_flow: cycles each wave spent WAITING (service waves 0..SVC-1: units not there yet / generation
order; stream waves: ring full) against the workgroup's total.  usage: python tools/gpu_flow_times.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hackrfdiags_amd import api
BLK = 262144
NSVC = int(os.environ.get("HRFD_NSVC", "6")); NSTREAM = 16 - NSVC
C, B = int(os.environ.get('HRFD_C', '256')), 16
dev = torch.device("cuda:0")
x = torch.randint(-128, 128, (C, B, BLK), dtype=torch.int8, device=dev)
pcm = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
torch.cuda.synchronize()
rx = api.Rx(C); rx.set_mode({'wbfm': api.WBFM, 'am': api.AM, 'fm': api.FM, 'ssb': api.LSB}[os.environ.get('HRFD_MODE', 'wbfm')])
grid = max(256, 8 * ((C + 7) // 8))
for _ in range(100):
    rx.process_device(x.data_ptr(), B * BLK, BLK, B, pcm.data_ptr())
rx.sync()
rx.debug_stamps(0); rx.debug_stamps(grid)
N = 24
rx.debug_enable_timing(N)
for _ in range(N):
    rx.process_device(x.data_ptr(), B * BLK, BLK, B, pcm.data_ptr())
rx.sync()
st = rx.debug_stamps(grid, read=True).astype(np.int64)
ms = [rx.debug_kernel_ms(i) for i in range(N)]
print("kernel ms:", " ".join(f"{m:.3f}" for m in ms))
print(f"last kernel {ms[-1]:.3f} ms; shader clock = {st[:, 0].mean() / ms[-1] / 1e3:.0f} MHz")
print("total cycles per workgroup: mean %.0f  min %d  max %d" % (st[:, 0].mean(), st[:, 0].min(), st[:, 0].max()))
print("waited / total per wave:", np.round(st[:, 8:24].mean(axis=0) / st[:, 0].mean(), 3).tolist())
pr = st[:, 24:32].mean(axis=0)
if pr.sum() > 0:
    # the stamp rows accumulate over the N launches of the burst
    print("probe, cycles per stream wave per launch: loop-top %d | grab+issue qb %d | carry (waits c16, qa) %d | ring wait %d | piece 0 %d | issue next %d | piece 1 %d | publish %d"
          % tuple((pr / N / NSTREAM).tolist()))
    print("   sum %.0f" % (pr.sum() / N / NSTREAM))
sp = st[:, 32:42].mean(axis=0)
if sp.sum() > 0:
    names = ["loop-top", "wait units", "patch", "partial sums", "wait P(g-1)", "seed+warm+tile", "wait order", "verify+U0+store", "D12+D40", "chk+publish"]
    print("service probe, cycles per service wave per launch (16 generations each):")
    print("   " + " | ".join(f"{n} {v:.0f}" for n, v in zip(names, (sp / N / NSVC).tolist())))
    print("   sum %.0f" % (sp.sum() / N / NSVC))
tl = st[:, 42:47]
if tl[:, 0].min() > 0:
    t0 = tl[:, 0].min()
    us = (tl - t0) / 100.0
    print("timeline of the LAST launch, us from the first workgroup's entry (100 MHz clock), over %d workgroups:" % len(tl))
    for i, n in enumerate(["entry", "tables loaded", "stream waves through", "service waves through", "last wave at the end"]):
        print("   %-24s min %7.2f  mean %7.2f  max %7.2f" % (n, us[:, i].min(), us[:, i].mean(), us[:, i].max()))
    print("   span first entry -> last end %.2f us; the events say %.2f us" % (us[:, 4].max(), 1e3 * ms[-1]))
    d = us[:, 4] - us[:, 0]
    print("   workgroup lifetime: min %.2f mean %.2f max %.2f us;  service tail behind the stream: mean %.2f max %.2f us"
          % (d.min(), d.mean(), d.max(), (us[:, 3] - us[:, 2]).mean(), (us[:, 3] - us[:, 2]).max()))
if tl[:, 0].min() > 0:
    # who is late?  workgroup ids go round the 8 XCDs
    st_us = (tl[:, 2] - tl[:, 0]) / 100.0
    print("stream time by XCD (blockIdx % 8): " + "  ".join("%d: %.1f" % (x, st_us[x::8].mean()) for x in range(8)))
    order = np.argsort(st_us)
    print("   slowest workgroups:", [(int(i), round(float(st_us[i]), 1)) for i in order[-8:]], " fastest:", [(int(i), round(float(st_us[i]), 1)) for i in order[:4]])
