# a long differential hunt: hundreds of randomized walks of calls against the oracle (receive side and modulators), shipped build
# (pytest writes into the log as it goes: a silent run is taken for hung)
set -e -o pipefail
mkdir -p gpurun_out
export HRFD_DEBUG_HOOKS=1
L=gpurun_out/r${ROUND:-5}_long_hunt.log
echo "# round ${ROUND:-5}, device code $(python3 -c 'import bench; print(bench.kernel_code_tag())'): randomized walks of calls against the oracle, shipped build" > $L
echo "# receive side, HRFD_WALK_SEEDS=${RX_SEEDS:-400}" >> $L
HRFD_WALK_SEEDS=${RX_SEEDS:-400} timeout -k 10 1000 python3 -u -m pytest tests/test_gpu_rx.py -q -m gpu -x -k random_walk >> $L 2>&1
echo "# modulators, HRFD_WALK_SEEDS=${TX_SEEDS:-300}" >> $L
HRFD_WALK_SEEDS=${TX_SEEDS:-300} timeout -k 10 300 python3 -u -m pytest tests/test_gpu_tx_nco.py -q -m gpu -x -k random_walk >> $L 2>&1
tail -12 $L
