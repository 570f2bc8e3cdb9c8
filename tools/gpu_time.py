"""Kernel time (HIP events) of the default bench workload on the in-tree libhrfd.so, a few repeats."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.gpu_ab import CHILD
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True)
    print(r.stdout.strip(), r.stderr.strip()[-300:] if r.returncode else "", flush=True)
