"""WBFM modulator: the last pass as one kernel (k_wb_tail, round 6) against rounds 2-5's two (k_wb_rails + k_mod<WB_TAIL>),
alternating in ONE process on one box (hook hrfd_mod_debug_set_tail), region time of back-to-back calls, output digests.
usage: python tools/wbmod_ab.py [C] [B] [reps] [rounds]"""
import os, sys, zlib
os.environ["HRFD_DEBUG_HOOKS"] = "1"
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hackrfdiags_amd import api
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(7)
n = 512 * B
pcm = torch.randint(-32768, 32768, (C, n), dtype=torch.int16, device=dev, generator=g)
out = torch.zeros((C, 512 * n), dtype=torch.int8, device=dev)
torch.cuda.synchronize()
st = torch.cuda.Stream(device=dev)
mods = {}
for tail in (0, 1):
    m = api.Mod(api.MOD_WBFM, C); m.debug_set_tail(tail); mods[tail] = m
by = C * n * 514
for r in range(rounds):
    for tail in (0, 1):
        m = mods[tail]
        for _ in range(max(3, reps // 4)):
            m.process_device(pcm.data_ptr(), n, out.data_ptr(), stream=st.cuda_stream)
        m.sync(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            m.process_device(pcm.data_ptr(), n, out.data_ptr(), stream=st.cuda_stream)
        e1.record(st); st.synchronize(); m.sync()
        ms = e0.elapsed_time(e1) / reps
        crc = zlib.crc32(out[:, : 1 << 15].cpu().numpy().tobytes() + out[:, -(1 << 15):].cpu().numpy().tobytes())
        print(f"wbfm {C}x{B} tail={'k_wb_tail' if tail else 'rails+cascade'}: {ms:.4f} ms per call = {by / ms / 1e6 / 8000:.4f} of 8 TB/s; crc {crc:08x}", flush=True)
