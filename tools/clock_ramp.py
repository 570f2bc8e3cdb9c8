"""How long does the clock governor take to settle under the headline's load?  Consecutive regions of 20 launches (one
HIP event pair each, no gaps between regions) from a cold start: ms per launch per region.  usage: python tools/clock_ramp.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hackrfdiags_amd import api
from hackrfdiags_amd.synth_torch import make_fm_batch
C, B, BLK = 256, 16, 262144
dev = torch.device("cuda:0")
x = make_fm_batch(C, B, dev)
pcm = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
torch.cuda.synchronize()
rx = api.Rx(C); rx.set_mode(api.WBFM)
s = torch.cuda.Stream()
time.sleep(1.0)
R, K = 60, 20
ev = [torch.cuda.Event(enable_timing=True) for _ in range(R + 1)]
ev[0].record(s)
for r in range(R):
    for _ in range(K):
        rx.process_device(x.data_ptr(), B * BLK, BLK, B, pcm.data_ptr(), stream=s.cuda_stream)
    ev[r + 1].record(s)
s.synchronize(); rx.sync()
ms = [ev[r].elapsed_time(ev[r + 1]) / K for r in range(R)]
print("ms per launch, regions of %d launches from a cold start:" % K)
print(" ".join(f"{m:.4f}" for m in ms))
