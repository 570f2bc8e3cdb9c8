#!/bin/bash
# the whole GPU suite against the shipped build, then against the stress build (-DHRFD_FLOW_CHAOS: every wave dawdles
# at random behind every hand-over point), progress into gpurun_out/ as it goes
set -e -o pipefail
mkdir -p gpurun_out
export HRFD_DEBUG_HOOKS=1
L=gpurun_out/r${ROUND:-5}_suite_${TAG:-a}.log
echo "# device code $(python3 -c 'import bench; print(bench.kernel_code_tag())'): the whole GPU suite, shipped build" > $L
timeout -k 10 1100 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -6 >> $L
echo "# the stress build (-DHRFD_FLOW_CHAOS)" >> $L
HRFD_LIB=$PWD/hackrfdiags_amd/lib/variants/chaos/libhrfd.so timeout -k 10 1100 python3 -m pytest tests -q -m gpu -x --deselect tests/test_dropin.py --deselect tests/test_shim.py 2>&1 | tail -6 >> $L
cat $L
