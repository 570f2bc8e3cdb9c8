#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
./tools/ubench/anyorder 2>&1 | tee gpurun_out/r5_anyorder.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_tx_nco.py tests/test_gpu_tools.py tests/test_count_raw.py tests/test_shim.py tests/test_dropin.py -q -m gpu -x 2>&1 | tail -8
