#!/bin/bash
# A/B on one box: the headline with extra bytes between the channels' input buffers (bench.py --stride-pad).
set -e -o pipefail
mkdir -p gpurun_out
out=gpurun_out/r5_stride_ab.txt
: > $out
for rep in 1 2 3; do
  for pad in 0 4096 8192 12288 36864 69632 135168 262144 1052672; do
    python3 bench.py --no-extras --no-cpu --steps 100 --warmup 100 --stride-pad $pad > gpurun_out/_line.json
    python3 - "$pad" <<'PY' >> $out
import json, sys
l = json.load(open("gpurun_out/_line.json"))
print("pad", sys.argv[1], "ms_per_step", l["ms_per_step"], "kernel_ms", l["roofline"]["kernel_ms_mean"], "frac", l["roofline"]["frac"])
PY
  done
done
cat $out
