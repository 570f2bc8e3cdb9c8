set -o pipefail
mkdir -p gpurun_out
export HRFD_DEBUG_HOOKS=1
L=gpurun_out/r${ROUND:-6}_final_suite.log
{
echo "# round ${ROUND:-6}, final device code ($(python3 -c 'import bench; print(bench.kernel_code_tag())')): the whole GPU suite against the stress build"
echo "# (-DHRFD_FLOW_CHAOS: every wave of the flow kernels and every mover of k_phase_scan dawdles at random behind its hand-overs;"
echo "#  tests/test_dropin.py and tests/test_shim.py link the shipped library by path, tests/test_gpu_hooks_off.py is the shipped state by definition: deselected)"
HRFD_LIB=$PWD/hackrfdiags_amd/lib/variants/chaos/libhrfd.so timeout -k 10 900 python3 -m pytest tests -q -m gpu --deselect tests/test_dropin.py --deselect tests/test_shim.py --deselect tests/test_gpu_hooks_off.py 2>&1 | tail -6
echo
echo "# random walks of calls against the oracle, HRFD_WALK_SEEDS=100 (receive side) / 60 (modulators), shipped build"
HRFD_WALK_SEEDS=100 timeout -k 10 900 python3 -m pytest tests/test_gpu_rx.py -q -m gpu -k random_walk 2>&1 | tail -3
HRFD_WALK_SEEDS=60 timeout -k 10 900 python3 -m pytest tests/test_gpu_tx_nco.py -q -m gpu -k random_walk 2>&1 | tail -3
echo
echo "# the same walks, 40 / 24 seeds, stress build"
HRFD_LIB=$PWD/hackrfdiags_amd/lib/variants/chaos/libhrfd.so HRFD_WALK_SEEDS=40 timeout -k 10 900 python3 -m pytest tests/test_gpu_rx.py -q -m gpu -k random_walk 2>&1 | tail -3
HRFD_LIB=$PWD/hackrfdiags_amd/lib/variants/chaos/libhrfd.so HRFD_WALK_SEEDS=24 timeout -k 10 900 python3 -m pytest tests/test_gpu_tx_nco.py -q -m gpu -k random_walk 2>&1 | tail -3
echo
echo "# bench.py --gpus 2 REHEARSED over gloo on this ONE GPU (not a measurement): the N > 1 code path of the round's bench.py"
HRFD_BENCH_REHEARSE=1 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r${ROUND:-6}_rehearsal_gloo_n2_NOT_A_MEASUREMENT.json 2> gpurun_out/r${ROUND:-6}_rehearsal.err; echo "rehearsal rc $?"; wc -c gpurun_out/r${ROUND:-6}_rehearsal_gloo_n2_NOT_A_MEASUREMENT.json
} > $L 2>&1
cat $L
