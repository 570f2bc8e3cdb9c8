#!/bin/bash
# timeline of the flow kernel's FIR modes and of WBFM on one box (probe build): where a workgroup's time goes
mkdir -p gpurun_out
export HRFD_DEBUG_HOOKS=1
export HRFD_LIB=$PWD/hackrfdiags_amd/lib/variants/probe/libhrfd.so
: > gpurun_out/r5_fir_times.txt
for m in wbfm am ssb fm; do
  nsvc=4; [ $m = wbfm ] && nsvc=6
  echo "== $m" >> gpurun_out/r5_fir_times.txt
  HRFD_MODE=$m HRFD_NSVC=$nsvc timeout -k 10 200 python3 tools/gpu_flow_times.py 2>/dev/null | grep -v "^kernel ms" >> gpurun_out/r5_fir_times.txt
done
cat gpurun_out/r5_fir_times.txt
