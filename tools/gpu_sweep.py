"""Sweep a tuning knob and print kernel time (HIP events)."""
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
rx.debug_enable_timing(8)
for stag in [4] + [4 + 256 * f for f in (1, 2, 4, 6, 16, 16 + 8, 16 + 8 + 6)]:
    rx.debug_set_stagger(stag)
    ts = []
    for rep in range(2):
        for i in range(8):
            rx.process_device(x.data_ptr(), B * BLK, BLK, B, pcm.data_ptr())
        rx.sync()
        ts += [rx.debug_kernel_ms(i) for i in range(8)]
    ts = ts[8:]
    print(f"flags {stag >> 8:2d}: kernel ms min {min(ts):.4f} mean {np.mean(ts):.4f} -> {C*B*BLK/np.mean(ts)/1e6:.0f} GB/s")
