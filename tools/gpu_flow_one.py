"""one shape through one kernel, against the oracle: python tools/gpu_flow_one.py bb C B run_len kern"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hackrfdiags_amd import api, synth
from tests.reflib import Oracle, WBFM
bb, C, B, run_len, kern = [int(a) for a in sys.argv[1:6]]
BLK = synth.BLOCK_BYTES
raw = np.concatenate([synth.make_input("fmtone" if c % 2 else "lcg", 120 + c, (B * bb + BLK - 1) // BLK)[: B * bb]
                      for c in range(C)]).reshape(C, B, bb)
rx = api.Rx(C); rx.set_mode(api.WBFM); rx.debug_set_stream(kern); rx.debug_set_run_len(run_len)
grid = 8 * ((C + 7) // 8) * B
rx.debug_stamps(grid)
print("launch", flush=True)
g = rx.process_block(raw, B)
print("done", rx.debug_counters(), flush=True)
st = rx.debug_stamps(grid, read=True)
for w in np.nonzero(st[:, 7])[0]:
    print("   dbg", [int(v) for v in st[w, 1:6]])
    print(f"  workgroup {w}: wait {int(st[w, 7]) >> 32} expired in wave {(int(st[w, 7]) & 0xffffffff) - 1}")
o = Oracle().rx(); o.set_mode(WBFM)
ok = True
for b in range(B):
    p, m, _, _ = o.process(raw[0, b])
    ok = ok and (g[0][0, b, :len(p)] == p).all() and int(g[2][0, b]) == m
print("oracle ok:", ok)
