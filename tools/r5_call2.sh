#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests/test_gpu_north_star_sizes.py tests/test_gpu_rx.py::test_full_size_bench_batch_matches_oracle tests/test_fanout.py tests/test_shard_gloo.py -q -m gpu -x -k "not soak" --durations=12 2>&1 | tee gpurun_out/r5_call2_tests.log | tail -25
python3 bench.py > gpurun_out/r5_bench_default_a.json 2> gpurun_out/r5_bench_default_a.err
tail -c 1500 gpurun_out/r5_bench_default_a.json
HRFD_BENCH_REHEARSE=1 HRFD_BENCH_SETTLE=0 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 > gpurun_out/r5_rehearsal_gloo_n2.json 2> gpurun_out/r5_rehearsal_gloo_n2.err
tail -c 3000 gpurun_out/r5_rehearsal_gloo_n2.json
