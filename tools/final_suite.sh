set -e -o pipefail
mkdir -p gpurun_out
export HRFD_DEBUG_HOOKS=1
L=gpurun_out/r${ROUND:-5}_final_suite.log
{
echo "# round ${ROUND:-5}, final device code ($(python3 -c 'import bench; print(bench.kernel_code_tag())')): the whole GPU suite against the shipped build"
timeout -k 10 900 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -8
echo
echo "# the whole GPU suite against the stress build (-DHRFD_FLOW_CHAOS; tests/test_dropin.py and tests/test_shim.py link the shipped library by path: deselected)"
HRFD_LIB=$PWD/hackrfdiags_amd/lib/variants/chaos/libhrfd.so timeout -k 10 900 python3 -m pytest tests -q -m gpu -x --deselect tests/test_dropin.py --deselect tests/test_shim.py 2>&1 | tail -8
echo
echo "# random walks of calls against the oracle, HRFD_WALK_SEEDS=100 (receive side) / 60 (modulators), shipped build"
HRFD_WALK_SEEDS=100 timeout -k 10 900 python3 -m pytest tests/test_gpu_rx.py -q -m gpu -x -k random_walk 2>&1 | tail -3
HRFD_WALK_SEEDS=60 timeout -k 10 900 python3 -m pytest tests/test_gpu_tx_nco.py -q -m gpu -x -k random_walk 2>&1 | tail -3
echo
echo "# the same walks, 40 / 24 seeds, stress build"
HRFD_LIB=$PWD/hackrfdiags_amd/lib/variants/chaos/libhrfd.so HRFD_WALK_SEEDS=40 timeout -k 10 900 python3 -m pytest tests/test_gpu_rx.py -q -m gpu -x -k random_walk 2>&1 | tail -3
HRFD_LIB=$PWD/hackrfdiags_amd/lib/variants/chaos/libhrfd.so HRFD_WALK_SEEDS=24 timeout -k 10 900 python3 -m pytest tests/test_gpu_tx_nco.py -q -m gpu -x -k random_walk 2>&1 | tail -3
echo
echo "# long soaks, HRFD_SOAK_LAUNCHES=3000, stress build then shipped build"
HRFD_LIB=$PWD/hackrfdiags_amd/lib/variants/chaos/libhrfd.so HRFD_SOAK_LAUNCHES=3000 timeout -k 10 900 python3 -m pytest tests/test_gpu_north_star_sizes.py -q -m gpu -x -k soak 2>&1 | tail -3
HRFD_SOAK_LAUNCHES=3000 timeout -k 10 900 python3 -m pytest tests/test_gpu_north_star_sizes.py -q -m gpu -x -k soak 2>&1 | tail -3
} > $L 2>&1
cat $L
