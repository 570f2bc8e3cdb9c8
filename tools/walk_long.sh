#!/bin/bash
# the walk of long batches (tests/test_gpu_rx.py::test_random_walk_of_long_batches) over many seeds against library variants
# usage: tools/walk_long.sh OUT "VARIANT:SEEDS" ...     (a failing variant does not stop the others)
set -o pipefail
mkdir -p gpurun_out
export HRFD_DEBUG_HOOKS=1
out=$1; shift
: > $out
for vs in "$@"; do
  v=${vs%%:*}; n=${vs##*:}
  lib=$PWD/hackrfdiags_amd/lib/variants/$v/libhrfd.so
  [ "$v" = ship ] && lib=$PWD/hackrfdiags_amd/lib/libhrfd.so
  echo "== $v, HRFD_WALK_SEEDS=$n" >> $out
  HRFD_LIB=$lib HRFD_WALK_SEEDS=$n timeout -k 10 900 python3 -u -m pytest tests/test_gpu_rx.py -q -m gpu -k random_walk_of_long_batches 2>&1 | grep -E "passed|failed|^FAILED|^E  " | head -40 >> $out
done
cat $out
