import sys, numpy as np, torch
sys.path.insert(0, '.')
import os
os.environ.setdefault("HRFD_DEBUG_HOOKS", "1")
from hackrfdiags_amd import api, synth
BLK = synth.BLOCK_BYTES
C, B = 5, 4
for kind in ("impulse", "zeros", "lcg"):
    xs = np.stack([synth.make_input(kind, 20 + c, 2 * B) for c in range(C)]).reshape(C, 2 * B, BLK)
    dev = torch.device("cuda:0")
    rx = api.Rx(C); rx.set_mode(api.WBFM)
    x = torch.from_numpy(xs[:, :B].copy()).to(dev)
    out = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
    torch.cuda.synchronize()
    rx.process_device(x.data_ptr(), B * BLK, BLK, B, out.data_ptr())
    print(kind, "sync", rx.sync(), "failed", rx.failed_channels(), "counters", rx.debug_counters())
