set -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r6_suite_final.log 2>&1; echo "suite rc $?"; tail -3 gpurun_out/r6_suite_final.log
python3 bench.py > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err; echo "bench rc $?"; wc -c gpurun_out/r6_bench_default.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_driver_shape.json 2> gpurun_out/r6_bench_driver_shape.err; echo "bench(driver shape) rc $?"
python3 bench.py --extras-verbose > gpurun_out/r6_bench_verbose.json 2> gpurun_out/r6_bench_verbose.err; echo "bench(verbose) rc $?"
bash tools/profile_round.sh r6 all > gpurun_out/r6_profile_round.log 2>&1; echo "profile rc $?"
ls gpurun_out/prof_r6 | head -60
