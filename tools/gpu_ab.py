"""A/B timing of tuning variants of libhrfd.so.

  python tools/gpu_ab.py build NAME "EXTRA FLAGS" ...   (on the build host)
  python tools/gpu_ab.py run [NAME ...]                 (on the GPU box)

`build` compiles hackrfdiags_amd/lib/variants/NAME/libhrfd.so with the extra
compiler flags; `run` times the WBFM kernel of each variant (own process, HIP
events) on the default bench workload and checks the PCM digest is identical.
"""
import os, subprocess, sys
os.environ.setdefault("HRFD_DEBUG_HOOKS", "1")   # the children use test hooks of include/hrfd_debug.h
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, "hackrfdiags_amd", "lib", "variants")

CHILD = r'''
import os, sys, zlib
import numpy as np
sys.path.insert(0, %r)
import torch
from hackrfdiags_amd import api
BLK = int(os.environ.get('HRFD_BLK', '262144'))
C, B = int(os.environ.get('HRFD_C', '256')), int(os.environ.get('HRFD_B', '16'))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randint(-128, 128, (C, B, BLK), dtype=torch.int8, device=dev, generator=g)
pcm = torch.zeros((C, B, 512), dtype=torch.int16, device=dev)
torch.cuda.synchronize()
rx = api.Rx(C); rx.set_mode(api.WBFM)
rx.debug_enable_timing(8)
if os.environ.get("HRFD_RUNLEN"):
    rx.debug_set_run_len(int(os.environ["HRFD_RUNLEN"]))
if os.environ.get("HRFD_STAGGER"):
    rx.debug_set_stagger(int(os.environ["HRFD_STAGGER"]))
if os.environ.get("HRFD_ATAN"):
    rx.debug_set_atan(int(os.environ["HRFD_ATAN"]))
if os.environ.get("HRFD_FLAGS"):
    rx.debug_set_stagger(4 + 256 * int(os.environ["HRFD_FLAGS"]))   # run-time ablation flags of old builds
# the clock governor needs ~25 ms of this load to settle: warm up first, then time 64 launches
for i in range(int(os.environ.get("HRFD_AB_WARM", "100"))):
    rx.process_device(x.data_ptr(), B * BLK, BLK, B, pcm.data_ptr())
rx.sync()
ts = []
for rep in range(8):
    for i in range(8):
        rx.process_device(x.data_ptr(), B * BLK, BLK, B, pcm.data_ptr())
    rx.sync()
    ts += [rx.debug_kernel_ms(i) for i in range(8)]
crc = zlib.crc32(pcm.cpu().numpy().tobytes())
print(f"kernel ms min {min(ts):.4f} mean {np.mean(ts):.4f} -> {C*B*BLK/np.mean(ts)/1e6:.0f} GB/s  pcm crc {crc:08x}")
''' % ROOT


def main():
    if sys.argv[1] == "build":
        args = sys.argv[2:]
        for name, extra in zip(args[0::2], args[1::2]):
            out = os.path.join(VDIR, name)
            os.makedirs(out, exist_ok=True)
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "hackrfdiags_amd", "csrc"),
                                   f"OUT={out}/libhrfd.so", f"OBJ={out}/hrfd_lib.o", f"EXTRA={extra}"])
    else:
        names = sys.argv[2:] or sorted(os.listdir(VDIR))
        for name in names:
            env = dict(os.environ, HRFD_LIB=os.path.join(VDIR, name, "libhrfd.so"))
            r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
            print(f"{name:12s} {r.stdout.strip()} {r.stderr.strip()[-300:] if r.returncode else ''}", flush=True)


if __name__ == "__main__":
    main()
