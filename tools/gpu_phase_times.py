"""Per-phase cycle breakdown of k_rx_wbfm<3> (s_memtime stamps of workgroup lane 0),
workgroup placement and phase-overlap statistics."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hackrfdiags_amd import api
BLK = 262144
C, B = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 16
stag = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
x = torch.randint(-128, 128, (C, B, BLK), dtype=torch.int8, device=dev)
pcm = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
torch.cuda.synchronize()
rx = api.Rx(C); rx.set_mode(api.WBFM); rx.debug_set_stagger(stag)
if os.environ.get('HRFD_WARM'):
    rx.debug_set_warm(int(os.environ['HRFD_WARM']))
RL = int(os.environ.get('HRFD_RUNLEN', '8'))
rx.debug_set_run_len(RL)
grid = 8 * ((C + 7) // 8) * ((B + RL - 1) // RL)   # one workgroup per run; the stamps are those of a run's LAST block
for _ in range(3):
    rx.process_device(x.data_ptr(), B * BLK, BLK, B, pcm.data_ptr()); rx.sync()
rx.debug_stamps(grid)
rx.debug_enable_timing(1)
rx.process_device(x.data_ptr(), B * BLK, BLK, B, pcm.data_ptr()); rx.sync()
st = rx.debug_stamps(grid, read=True)
ms = rx.debug_kernel_ms(0)
hw = (st[:, 6] & np.uint64(0xFFFFFFFF)).astype(np.int64); xcc = (st[:, 6] >> np.uint64(32)).astype(np.int64) & 0xF
st = st.astype(np.int64)
names = ["A: front end+atan2+v", "reduce+gate", "B: patch+seeds+chains+check", "-", "C: cvt+D8+D12+D40"]
d = np.diff(st[:, :6], axis=1)
print(f"kernel {ms:.3f} ms with stamps; grid {grid}; stagger {stag}")
for i, n in enumerate(names):
    print(f"  {n:28s} mean {d[:, i].mean():9.0f}  p50 {np.median(d[:, i]):9.0f}  max {d[:, i].max():9.0f} cycles")
print(f"  total per workgroup          mean {(st[:,5]-st[:,0]).mean():9.0f} cycles")
wa = st[:, 8:24] - st[:, 0:1]
print("  end of phase A per wave (mean cycles since block start):", np.round(wa.mean(axis=0)).astype(int).tolist())
print("  slowest wave index histogram:", np.bincount(np.argmax(wa, axis=1), minlength=16).tolist())
wb = st[:, 24:28] - st[:, 2:3]
print("  end of recurrence per B-wave (mean cycles since B start):", np.round(wb.mean(axis=0)).astype(int).tolist())
cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
key = xcc * 1000 + se * 100 + sh * 50 + cu
print("  distinct (xcc,se,sh,cu):", len(np.unique(key)), " xcc of WG0..15:", xcc[:16].tolist())
print("  cu-key of WG 0..23:", key[:24].tolist())
# co-residency: for the first 512 WGs, which pairs share a cu-key?
first = {}
pairs = []
for w in range(min(grid, 512)):
    k = key[w]
    if k in first: pairs.append((first[k], w))
    else: first[k] = w
print("  first co-resident pairs:", pairs[:12])
# overlap: per CU, fraction of time where at least one resident WG is in phase A
tot = 0; inA = 0; inB2 = 0
for k in np.unique(key):
    idx = np.nonzero(key == k)[0]
    t0, t1 = st[idx, 0].min(), st[idx, 5].max()
    ev = []
    for w in idx:
        ev.append((st[w, 0], +1, 0)); ev.append((st[w, 1], -1, 0))      # phase A interval
        ev.append((st[w, 2], +1, 1)); ev.append((st[w, 3], -1, 1))      # phase B interval
    ev.sort()
    nA = nB = 0; last = t0
    for t, dlt, which in ev:
        if nA > 0: inA += t - last
        if nA == 0 and nB > 0: inB2 += t - last
        last = t
        if which == 0: nA += dlt
        else: nB += dlt
    tot += t1 - t0
print(f"  per-CU time with >=1 WG in phase A: {100*inA/tot:.1f}%   only-B (no A resident): {100*inB2/tot:.1f}%")
# timeline of one CU
k0 = key[0]
idx = np.nonzero(key == k0)[0]
t0 = st[idx, 0].min()
print("  timeline of the CU that ran WG 0 (ticks/1000 since first start):  WG: A_start A_end(w0) B_start B_end C_end")
for w in idx[np.argsort(st[idx, 0])]:
    print("   WG %5d: %7.1f %7.1f %7.1f %7.1f %7.1f" % (w, (st[w,0]-t0)/1e3, (st[w,1]-t0)/1e3, (st[w,2]-t0)/1e3, (st[w,3]-t0)/1e3, (st[w,5]-t0)/1e3))
