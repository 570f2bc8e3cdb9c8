#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_rx.py -q -m gpu -x --durations=8 2>&1 | tee gpurun_out/r5_call3_tests.log | tail -15
bash tools/flow_ab.sh gpurun_out/r5_flow_split_ab_1.txt old ship svc5 svc7
