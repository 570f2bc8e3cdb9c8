#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
HRFD_LIB=$PWD/hackrfdiags_amd/lib/variants/probe/libhrfd.so python3 tools/gpu_flow_times.py > gpurun_out/r5_flow_times_split3.txt 2>&1
tail -16 gpurun_out/r5_flow_times_split3.txt
AB_VERIFY=4 bash tools/flow_ab.sh gpurun_out/r5_flow_split_ab_3.txt old ship
