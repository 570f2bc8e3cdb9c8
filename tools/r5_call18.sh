#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r5_fir_prio_ab.txt
for rep in 1 2 3; do
  for v in ship firprio0 firprio1; do
    lib=$PWD/hackrfdiags_amd/lib/variants/$v/libhrfd.so; [ "$v" = ship ] && lib=$PWD/hackrfdiags_amd/lib/libhrfd.so
    for wl in am fm ssb mixed; do
      HRFD_LIB=$lib python3 bench.py --workload $wl --no-cpu --no-extras --steps 100 --warmup 100 --verify 2 > gpurun_out/_line.json
      python3 - "$v" "$wl" >> gpurun_out/r5_fir_prio_ab.txt <<'PY'
import json, sys
l = json.load(open("gpurun_out/_line.json"))
print(f"{sys.argv[1]:10s} {sys.argv[2]:6s} 256x16 ms_per_step {l['ms_per_step']:.4f} frac {l['roofline']['frac']:.4f} oracle_ok {l['verification'].get('oracle_channels_checked')} uncommitted {l['verification']['uncommitted_launches']}")
PY
    done
  done
done
sort -k2,2 -k1,1 -s gpurun_out/r5_fir_prio_ab.txt
