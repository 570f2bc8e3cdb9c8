"""kernel time of the WBFM batch kernels at 256 / 512 / 1024 channels x 16 blocks (random IQ): python tools/gpu_flow_scale.py"""
import os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hackrfdiags_amd import api
BLK, B = 262144, 16
dev = torch.device("cuda:0")
for C in (256, 512, 1024):
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.randint(-128, 128, (C, B, BLK), dtype=torch.int8, device=dev, generator=g)
    pcm = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
    for kern in (2, 1):
        rx = api.Rx(C); rx.set_mode(api.WBFM); rx.debug_set_stream(kern)
        rx.debug_enable_timing(8)
        for i in range(40):
            rx.process_device(x.data_ptr(), B * BLK, BLK, B, pcm.data_ptr())
        rx.sync()
        ts = []
        for rep in range(4):
            for i in range(8):
                rx.process_device(x.data_ptr(), B * BLK, BLK, B, pcm.data_ptr())
            rx.sync()
            ts += [rx.debug_kernel_ms(i) for i in range(8)]
        crc = zlib.crc32(pcm.cpu().numpy().tobytes())
        print(f"C {C} kernel {kern}: ms mean {np.mean(ts):.4f} -> {C*B*(BLK+1028)/np.mean(ts)/1e6:.0f} GB/s ({C*B*(BLK+1028)/np.mean(ts)/8e9:.3f} of 8 TB/s) crc {crc:08x} counters {rx.debug_counters()}", flush=True)
    del x, pcm
