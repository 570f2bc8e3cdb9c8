"""Soak: N launches of the full-size WBFM batch (256 channels x 16 blocks) on fresh random and FM-like
input, through k_rx_wbfm_stream and through k_rx_wbfm (two handles, both carrying their channel state
from launch to launch).  The two kernels share the arithmetic but none of the scheduling (roles,
LDS counters, dynamic runs, quad layout vs. chunk layout), so any ordering bug in either shows up
as a PCM / magnitude difference.  usage: python tools/gpu_soak.py [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hackrfdiags_amd import api
BLK = 262144
C, B = 256, 16
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(2026)
a, b = api.Rx(C), api.Rx(C)
for rx in (a, b):
    rx.set_mode(api.WBFM)
b.debug_set_stream(False)
pa = torch.zeros((C, B, 512), dtype=torch.int16, device=dev); pb = torch.zeros_like(pa)
ma = torch.zeros((C, B), dtype=torch.int32, device=dev); mb = torch.zeros_like(ma)
bad = 0
for it in range(N):
    if it % 3 == 2:
        # slowly rotating phasor + noise: long runs of wrapping phase differences
        t = torch.arange(B * BLK // 2, device=dev, dtype=torch.float32)
        ph = t * (0.3 + 0.01 * it)
        x = torch.stack([(100 * torch.cos(ph)).to(torch.int8), (100 * torch.sin(ph)).to(torch.int8)], dim=1).reshape(1, B, BLK)
        x = (x + torch.randint(-3, 4, (C, B, BLK), dtype=torch.int8, device=dev, generator=g)).contiguous()
    else:
        x = torch.randint(-128, 128, (C, B, BLK), dtype=torch.int8, device=dev, generator=g)
    torch.cuda.synchronize()
    a.process_device(x.data_ptr(), B * BLK, BLK, B, pa.data_ptr(), d_magnitude=ma.data_ptr())
    b.process_device(x.data_ptr(), B * BLK, BLK, B, pb.data_ptr(), d_magnitude=mb.data_ptr())
    va, vb = a.sync(), b.sync()
    same = bool(torch.equal(pa, pb)) and bool(torch.equal(ma, mb))
    if not same or va or vb:
        bad += 1
        print(f"launch {it}: equal={same} violations stream={va} per-block={vb}", flush=True)
    if it % 25 == 24:
        print(f"{it + 1} launches, {bad} bad; counters stream {a.debug_counters()} per-block {b.debug_counters()}", flush=True)
print("SOAK", "OK" if bad == 0 else "FAILED", N, "launches")
sys.exit(1 if bad else 0)
