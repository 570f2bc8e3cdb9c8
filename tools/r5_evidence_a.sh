#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HRFD_DEBUG_HOOKS=1
timeout -k 10 700 python3 -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/r5_suite_b.log 2>&1; echo "suite rc $?"; tail -3 gpurun_out/r5_suite_b.log
unset HRFD_DEBUG_HOOKS
python3 bench.py > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err; echo "bench rc $?"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench_driver_shape.json 2> gpurun_out/r5_bench_driver_shape.err; echo "bench(driver shape) rc $?"
bash tools/profile_round.sh r5 all > gpurun_out/r5_profile_round.log 2>&1; echo "profile rc $?"
ls gpurun_out/prof_r5 | head -40
