#!/bin/bash
# One evidence round on a GPU box: the whole GPU suite, the default bench line, the driver-shaped line, rocprofv3 kernel stats of
# every workload (tools/profile_round.sh).  The counters are rounds of their own: tools/pmc_round.sh, tools/sq_round.sh.
set -o pipefail
mkdir -p gpurun_out
export HRFD_DEBUG_HOOKS=1
timeout -k 10 700 python3 -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/r${ROUND:-5}_suite_b.log 2>&1; echo "suite rc $?"; tail -3 gpurun_out/r${ROUND:-5}_suite_b.log
unset HRFD_DEBUG_HOOKS
python3 bench.py > gpurun_out/r${ROUND:-5}_bench_default.json 2> gpurun_out/r${ROUND:-5}_bench_default.err; echo "bench rc $?"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r${ROUND:-5}_bench_driver_shape.json 2> gpurun_out/r${ROUND:-5}_bench_driver_shape.err; echo "bench(driver shape) rc $?"
bash tools/profile_round.sh r${ROUND:-5} all > gpurun_out/r${ROUND:-5}_profile_round.log 2>&1; echo "profile rc $?"
ls gpurun_out/prof_r${ROUND:-5} | head -40
