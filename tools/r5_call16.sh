#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HRFD_DEBUG_HOOKS=1
HRFD_LIB=$PWD/hackrfdiags_amd/lib/variants/ring14_512/libhrfd.so timeout -k 10 400 python3 -m pytest tests/test_gpu_rx.py -q -m gpu -x -p no:cacheprovider > gpurun_out/r5_call16_tests.log 2>&1; echo "rc $?"; tail -3 gpurun_out/r5_call16_tests.log
unset HRFD_DEBUG_HOOKS
: > gpurun_out/r5_fir_ring_ab.txt
for rep in 1 2 3; do
  for v in ship ring14_512; do
    lib=$PWD/hackrfdiags_amd/lib/variants/$v/libhrfd.so; [ "$v" = ship ] && lib=$PWD/hackrfdiags_amd/lib/libhrfd.so
    for wl in am ssb mixed; do
      HRFD_LIB=$lib python3 bench.py --workload $wl --no-cpu --no-extras --steps 100 --warmup 100 --verify 2 > gpurun_out/_line.json
      python3 - "$v" "$wl" >> gpurun_out/r5_fir_ring_ab.txt <<'PY'
import json, sys
l = json.load(open("gpurun_out/_line.json"))
print(f"{sys.argv[1]:12s} {sys.argv[2]:6s} 256x16 ms_per_step {l['ms_per_step']:.4f} frac {l['roofline']['frac']:.4f} oracle_ok {l['verification'].get('oracle_channels_checked')} uncommitted {l['verification']['uncommitted_launches']}")
PY
    done
  done
done
cat gpurun_out/r5_fir_ring_ab.txt
