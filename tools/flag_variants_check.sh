#!/bin/bash
# every build flag that is left OFF still gives the oracle's results (variants built by the caller: tools/gpu_ab.py build NAME "FLAGS")
# usage: tools/flag_variants_check.sh OUT NAME...
set -o pipefail
mkdir -p gpurun_out
export HRFD_DEBUG_HOOKS=1
out=$1; shift
: > $out
for v in "$@"; do
  lib=$PWD/hackrfdiags_amd/lib/variants/$v/libhrfd.so
  echo "== $v ($(cat hackrfdiags_amd/lib/variants/$v/FLAGS 2>/dev/null))" >> $out
  HRFD_LIB=$lib timeout -k 10 400 python3 -m pytest tests/test_gpu_rx.py tests/test_gpu_tx_nco.py -q -m gpu -x -k "full_size or long_runs or mixed_bank or fir_modes_on_the_flow or closed_gates or fm_modulator or golden_rx" 2>&1 | tail -2 >> $out
done
cat $out
