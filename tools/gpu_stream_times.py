"""Busy cycles per wave of k_rx_wbfm_stream (stream waves 4..15, service waves 0..3) against the
workgroup's total: who waits for whom at the per-block barrier."""
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
torch.cuda.synchronize()
rx = api.Rx(C); rx.set_mode(api.WBFM)
if os.environ.get('HRFD_RUNLEN'):
    rx.debug_set_run_len(int(os.environ['HRFD_RUNLEN']))
RL = int(os.environ.get('HRFD_RUNLEN', '16'))
grid = 8 * ((C + 7) // 8) * ((B + RL - 1) // RL)
for _ in range(3):
    rx.process_device(x.data_ptr(), B * BLK, BLK, B, pcm.data_ptr()); rx.sync()
rx.debug_stamps(grid)
N = int(os.environ.get('HRFD_BURST', '24'))
rx.debug_enable_timing(N)
for _ in range(N):
    rx.process_device(x.data_ptr(), B * BLK, BLK, B, pcm.data_ptr())
rx.sync()
st = rx.debug_stamps(grid, read=True).astype(np.int64)
ms = [rx.debug_kernel_ms(i) for i in range(N)]
print("kernel ms of the burst:", " ".join(f"{m:.3f}" for m in ms))
print(f"last kernel {ms[-1]:.3f} ms; grid {grid}; shader clock during it = {st[:, 0].mean() / ms[-1] / 1e3:.0f} MHz (cycles of a workgroup / kernel time)")
print("total cycles per workgroup: mean %.0f  min %d  max %d" % (st[:, 0].mean(), st[:, 0].min(), st[:, 0].max()))
print("busy cycles per wave (mean over workgroups):", np.round(st[:, 8:24].mean(axis=0)).astype(int).tolist())
print("busy / total:", np.round(st[:, 8:24].mean(axis=0) / st[:, 0].mean(), 2).tolist())
