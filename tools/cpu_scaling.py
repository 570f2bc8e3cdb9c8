#!/usr/bin/env python3
"""How the reference's CPU chain (oracle/_ref, ref_bench_rx: one IqDataProcessor per std::thread) scales with the
number of host threads on this box, and what the box says about its CPUs.  Output: one line per thread count."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hackrfdiags_amd import synth  # noqa: E402
from tests import reflib  # noqa: E402

print("host_core_counts (hw threads, physical cores, cgroup quota):", bench.host_core_counts(), "os.cpu_count", os.cpu_count())
for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
             "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/cpuset/cpuset.cpus"):
    try:
        print(path, "=", open(path).read().strip())
    except OSError as e:
        print(path, "-", e.strerror)
x = synth.make_input("fmtone", 0, 8).reshape(8, 262144)
eng = reflib.Ref()
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
base = None
for t in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    if t > (os.cpu_count() or 1):
        break
    n, dt, _ = eng.bench_rx(reflib.WBFM, t, secs, x)
    v = n * 131072 / dt / 1e6
    base = base or v
    print(f"threads {t:4d}: {v:9.1f} MSamples/s  x{v / base:6.1f}  per-thread efficiency {v / base / t:5.2f}", flush=True)
