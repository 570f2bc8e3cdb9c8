"""k_rx_wbfm_flow against k_rx_wbfm_stream and the oracle on a few shapes, then its kernel time on the bench shape.
usage (GPU box): python tools/gpu_flow_check.py [quick]"""
import os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hackrfdiags_amd import api, synth
from tests.reflib import Oracle, WBFM

BLK = synth.BLOCK_BYTES
orc = Oracle()
bad = 0
for bb, C, B in ((262144, 3, 6), (32768, 2, 9), (40960, 5, 7), (262144, 9, 17)):
    raw = np.concatenate([synth.make_input("fmtone" if c % 2 else "lcg", 120 + c, (B * bb + BLK - 1) // BLK)[: B * bb]
                          for c in range(C)]).reshape(C, B, bb)
    for run_len in (0, 2, 3):
        outs = {}
        for kern in (2, 1):
            rx = api.Rx(C); rx.set_mode(api.WBFM); rx.debug_set_stream(kern); rx.debug_set_run_len(run_len)
            h = B // 2
            g1 = rx.process_block(raw[:, :h], h)
            g2 = rx.process_block(raw[:, h:], B - h)
            outs[kern] = (np.concatenate([g1[0], g2[0]], axis=1), np.concatenate([g1[2], g2[2]], axis=1), rx.debug_counters())
        same = (outs[1][0] == outs[2][0]).all() and (outs[1][1] == outs[2][1]).all()
        ok_or = True
        for c in range(min(C, 2)):
            o = orc.rx(); o.set_mode(WBFM)
            for b in range(B):
                p, m, _, _ = o.process(raw[c, b])
                if not ((outs[2][0][c, b, :len(p)] == p).all() and int(outs[2][1][c, b]) == m):
                    ok_or = False
                    print("   oracle mismatch", bb, run_len, c, b, int((outs[2][0][c, b, :len(p)] != p).sum()), int(outs[2][1][c, b]), m)
        print(f"bb {bb} C {C} B {B} run_len {run_len}: flow == stream {same}; flow == oracle {ok_or}; counters flow {outs[2][2]} stream {outs[1][2]}", flush=True)
        bad += (not same) + (not ok_or) + (outs[2][2][5] != 0)
print("FAILURES:", bad, flush=True)
if len(sys.argv) > 1:
    sys.exit(1 if bad else 0)
# timing on the bench shape
C, B = int(os.environ.get('HRFD_C', '256')), 16
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randint(-128, 128, (C, B, BLK), dtype=torch.int8, device=dev, generator=g)
for kern in (2, 1):
    pcm = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
    rx = api.Rx(C); rx.set_mode(api.WBFM); rx.debug_set_stream(kern)
    rx.debug_enable_timing(8)
    for i in range(100):
        rx.process_device(x.data_ptr(), B * BLK, BLK, B, pcm.data_ptr())
    rx.sync()
    ts = []
    for rep in range(8):
        for i in range(8):
            rx.process_device(x.data_ptr(), B * BLK, BLK, B, pcm.data_ptr())
        rx.sync()
        ts += [rx.debug_kernel_ms(i) for i in range(8)]
    crc = zlib.crc32(pcm.cpu().numpy().tobytes())
    print(f"kernel {kern}: ms min {min(ts):.4f} mean {np.mean(ts):.4f} -> {C*B*BLK/np.mean(ts)/1e6:.0f} GB/s  pcm crc {crc:08x} counters {rx.debug_counters()}", flush=True)
sys.exit(1 if bad else 0)
