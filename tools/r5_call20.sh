#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r5_iqdump_ab.txt
for rep in 1 2 3; do
  for v in mid ship old; do
    lib=$PWD/hackrfdiags_amd/lib/variants/$v/libhrfd.so; [ "$v" = ship ] && lib=$PWD/hackrfdiags_amd/lib/libhrfd.so
    HRFD_LIB=$lib python3 bench.py --iqdump --no-cpu --no-extras --steps 100 --warmup 100 --verify 0 > gpurun_out/_line.json
    python3 - "$v" >> gpurun_out/r5_iqdump_ab.txt <<'PY'
import json, sys
l = json.load(open("gpurun_out/_line.json"))
print(f"{sys.argv[1]:6s} iqdump 256x16 ms_per_step {l['ms_per_step']:.4f} frac {l['roofline']['frac']:.4f} uncommitted {l['verification']['uncommitted_launches']}")
PY
  done
done
cat gpurun_out/r5_iqdump_ab.txt
