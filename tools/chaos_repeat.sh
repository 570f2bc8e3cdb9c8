#!/bin/bash
# one test of the GPU suite N times against library variants (a timing-dependent failure under the stress build:
# how often before a change, how often after).  usage: tools/chaos_repeat.sh OUT N "pytest selection" VARIANT...
set -o pipefail
mkdir -p gpurun_out
export HRFD_DEBUG_HOOKS=1
out=$1; n=$2; sel=$3; shift 3
: > $out
for v in "$@"; do
  lib=$PWD/hackrfdiags_amd/lib/variants/$v/libhrfd.so
  [ "$v" = ship ] && lib=$PWD/hackrfdiags_amd/lib/libhrfd.so
  ok=0; bad=0
  for i in $(seq $n); do
    if HRFD_LIB=$lib timeout -k 10 300 python3 -m pytest $sel -q -m gpu -x > gpurun_out/_rep.log 2>&1; then ok=$((ok + 1)); else bad=$((bad + 1)); grep -E "^E  |^FAILED" gpurun_out/_rep.log | head -4 >> $out; fi
    echo "$v run $i: passed $ok failed $bad" >> $out
  done
  echo "== $v: $ok passed, $bad failed of $n  ($sel)" >> $out
done
cat $out
