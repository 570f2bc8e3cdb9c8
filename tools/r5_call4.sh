#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
HRFD_LIB=$PWD/hackrfdiags_amd/lib/variants/probe/libhrfd.so python3 tools/gpu_flow_times.py > gpurun_out/r5_flow_times_split2.txt 2>&1
cat gpurun_out/r5_flow_times_split2.txt | tail -22
bash tools/flow_ab.sh gpurun_out/r5_flow_split_ab_2.txt old ship svc5 svc4
