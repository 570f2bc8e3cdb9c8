#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HRFD_DEBUG_HOOKS=1
timeout -k 10 400 python3 -m pytest tests/test_gpu_rx.py tests/test_gpu_north_star_sizes.py tests/test_fanout.py tests/test_gpu_ingest.py -q -m gpu -x -k "not soak" -p no:cacheprovider > gpurun_out/r5_call17_tests.log 2>&1; echo "rc $?"; tail -3 gpurun_out/r5_call17_tests.log
HRFD_LIB=$PWD/hackrfdiags_amd/lib/variants/chaos/libhrfd.so timeout -k 10 500 python3 -m pytest tests/test_gpu_rx.py tests/test_gpu_north_star_sizes.py -q -m gpu -x -p no:cacheprovider > gpurun_out/r5_call17_chaos.log 2>&1; echo "rc $?"; tail -3 gpurun_out/r5_call17_chaos.log
HRFD_LIB=$PWD/hackrfdiags_amd/lib/variants/probe/libhrfd.so timeout -k 10 120 python3 tools/gpu_flow_times.py > gpurun_out/r5_flow_times_lt.txt 2>&1
tail -9 gpurun_out/r5_flow_times_lt.txt
unset HRFD_DEBUG_HOOKS
AB_VERIFY=2 timeout -k 10 600 bash tools/flow_ab.sh gpurun_out/r5_flow_lasttheta_ab.txt nolt ship
