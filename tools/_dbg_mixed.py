import sys, os, numpy as np, torch
sys.path.insert(0, '.')
os.environ.setdefault("HRFD_DEBUG_HOOKS", "1")
from hackrfdiags_amd import api, synth
from tests.reflib import WBFM
BLK = synth.BLOCK_BYTES
C, B = int(sys.argv[1]), int(sys.argv[2])
xs = np.stack([synth.make_input("lcg" if c % 2 else "amtone", 60 + c, B) for c in range(C)]).reshape(C, B, BLK)
dev = torch.device("cuda:0")
rx = api.Rx(C); rx.set_mode(api.WBFM)
x = torch.from_numpy(xs).to(dev)
out = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
torch.cuda.synchronize()
print("launch", C, B, flush=True)
rx.process_device(x.data_ptr(), B * BLK, BLK, B, out.data_ptr())
print("sync", rx.sync(), "failed", rx.failed_channels(), "counters", rx.debug_counters(), flush=True)
