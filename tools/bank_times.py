"""k_rx_flow_bank on the probe build: a workgroup's lifetime by kind and by XCD (blockIdx % 8) for the mixed bank of
BASELINE config 3 (64 AM + 64 FM + 64 WBFM + 64 SSB, 16 blocks).  usage: HRFD_LIB=.../variants/probe/libhrfd.so python tools/bank_times.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hackrfdiags_amd import api
BLK = 262144
C, B = 256, 16
dev = torch.device("cuda:0")
x = torch.randint(-128, 128, (C, B, BLK), dtype=torch.int8, device=dev)
pcm = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
kinds = [api.AM, api.FM, api.WBFM, api.LSB]
names = {api.AM: "AM", api.FM: "FM", api.WBFM: "WBFM", api.LSB: "SSB"}
order = os.environ.get("HRFD_BANK_ORDER", "blocks")      # blocks: 64 of a kind after another; interleaved: c % 4
mode = [kinds[c // 64] if order == "blocks" else kinds[c % 4] for c in range(C)]
rx = api.Rx(C)
for c in range(C):
    rx.set_mode(mode[c], channel=c)
for _ in range(100):
    rx.process_device(x.data_ptr(), B * BLK, BLK, B, pcm.data_ptr())
rx.sync()
rx.debug_stamps(0); rx.debug_stamps(256)
N = 24
rx.debug_enable_timing(N)
for _ in range(N):
    rx.process_device(x.data_ptr(), B * BLK, BLK, B, pcm.data_ptr())
rx.sync()
st = rx.debug_stamps(256, read=True).astype(np.int64)
ms = [rx.debug_kernel_ms(i) for i in range(N)]
print("order %s; kernel ms: mean %.4f min %.4f" % (order, float(np.mean(ms)), float(np.min(ms))))
tl = st[:, 42:47]
t0 = tl[:, 0].min()
life = (tl[:, 4] - t0) / 100.0                           # us from the first entry to the workgroup's end (last launch)
print("end of the last launch's workgroups, us from the first entry: mean %.1f max %.1f; the events say %.1f" % (life.mean(), life.max(), 1e3 * ms[-1]))
# which channel a workgroup runs: the library's list 9 (hrfd_api.hip: WBFM on the even positions; HRFD_BANK_XCD=0: channel order)
if os.environ.get("HRFD_BANK_XCD", "1") == "1":
    heavy = [c for c in range(C) if mode[c] == api.WBFM]; light = [c for c in range(C) if mode[c] != api.WBFM]
    lst = []
    for p_ in range(C):
        take_heavy = (p_ % 2 == 0 and heavy) or not light
        lst.append(heavy.pop(0) if take_heavy else light.pop(0))
    mode = [mode[c] for c in lst]
for k in kinds:
    sel = np.array([mode[w] == k for w in range(C)])
    print("  %-4s end mean %.1f max %.1f | even XCDs %.1f odd XCDs %.1f" % (names[k], life[sel].mean(), life[sel].max(),
          life[sel & (np.arange(C) % 2 == 0)].mean(), life[sel & (np.arange(C) % 2 == 1)].mean()))
print("  by XCD: " + "  ".join("%d: %.1f" % (q, life[q::8].mean()) for q in range(8)))
