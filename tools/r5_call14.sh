#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HRFD_DEBUG_HOOKS=1
timeout -k 10 600 python3 -m pytest tests/test_gpu_tx_nco.py tests/test_count_raw.py tests/test_shim.py tests/test_dropin.py tests/test_gpu_tools.py -q -m gpu -x -p no:cacheprovider > gpurun_out/r5_call14_tests.log 2>&1; echo "rc $?"; tail -3 gpurun_out/r5_call14_tests.log
unset HRFD_DEBUG_HOOKS
for rep in 1 2 3; do
  python3 bench.py --workload fmmod --no-cpu --no-extras --steps 60 --warmup 40 > gpurun_out/_line.json
  python3 -c "import json; l=json.load(open('gpurun_out/_line.json')); print('fmmod 1024x16 ms_per_step', l['ms_per_step'], 'frac', l['roofline']['frac'])"
done
