"""Times one modulator bank of the library HRFD_LIB points at (default: the shipped one): region time of `reps`
back-to-back calls behind `warm` untimed ones, one HIP event pair on the launch stream, and a digest of the output.
usage: python tools/mod_time.py KIND [C] [B] [reps] [warm]   KIND: ssb | interp | am | fm | wbfm
(A/B: alternate `HRFD_LIB=.../variants/NAME/libhrfd.so python tools/mod_time.py ...` on ONE box: tools/mod_ab.sh.)"""
import os, sys, zlib
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
if os.environ.get("HRFD_MOD_TAIL") or os.environ.get("HRFD_MOD_SCAN") or os.environ.get("HRFD_MOD_SLICED"):
    os.environ["HRFD_DEBUG_HOOKS"] = "1"               # (read once by the library: before it is loaded)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hackrfdiags_amd import api
kind = sys.argv[1]
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 100
warm = int(sys.argv[5]) if len(sys.argv) > 5 else 60
K = {"ssb": api.MOD_SSB, "interp": api.MOD_INTERP, "am": api.MOD_AM, "fm": api.MOD_FM, "wbfm": api.MOD_WBFM}[kind]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(7)
amp = int(os.environ.get("HRFD_MOD_AMP", "32768"))
n = 512 * B
pcm = torch.randint(-amp, amp, (C, n * (2 if kind == "interp" else 1)), dtype=torch.int16, device=dev, generator=g)
out = torch.zeros((C, 512 * n), dtype=torch.int8, device=dev)
torch.cuda.synchronize()
m = api.Mod(K, C)
if os.environ.get("HRFD_MOD_TAIL"):
    m.debug_set_tail(int(os.environ["HRFD_MOD_TAIL"]))   # WBFM: 0 = k_wb_rails + k_mod<WB_TAIL>, 1 = k_wb_tail
if os.environ.get("HRFD_MOD_SCAN"):
    m.debug_set_scan(int(os.environ["HRFD_MOD_SCAN"]))   # FM / WBFM: 0 = k_phase_rows8 / rows, 1 = k_phase_scan<64>, 2 = k_phase_rows
if os.environ.get("HRFD_MOD_SLICED"):
    m.debug_set_sliced(int(os.environ["HRFD_MOD_SLICED"]))   # FM / WBFM: 0 = one pass after the other, 1 = automatic, 2 = always sliced
st = torch.cuda.Stream(device=dev)
for _ in range(warm):
    m.process_device(pcm.data_ptr(), n, out.data_ptr(), stream=st.cuda_stream)
m.sync(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(reps):
    m.process_device(pcm.data_ptr(), n, out.data_ptr(), stream=st.cuda_stream)
e1.record(st); st.synchronize(); m.sync()
ms = e0.elapsed_time(e1) / reps
# digest of a slice of every channel (the whole output is 4 GiB at 1024 x 16)
sl = out[:, : 1 << 16].cpu().numpy().tobytes() + out[:, -(1 << 16):].cpu().numpy().tobytes()
by = C * n * 514
print(f"{kind} {C}x{B}: {ms:.4f} ms per call, {by / ms / 1e6:.0f} GB/s = {by / ms / 1e6 / 8000:.4f} of 8 TB/s; crc {zlib.crc32(sl):08x}", flush=True)
