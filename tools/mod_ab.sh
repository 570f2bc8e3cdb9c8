#!/bin/bash
# Alternating A/B of library variants on a modulator bank, on the GPU box: tools/mod_ab.sh "KIND C B" ROUNDS NAME [NAME ...]
# (NAME: a directory under hackrfdiags_amd/lib/variants built by `python tools/gpu_ab.py build NAME "FLAGS"`, or "shipped")
W=$1; R=$2; shift 2
for r in $(seq $R); do
  for v in "$@"; do
    if [ "$v" = shipped ]; then L=hackrfdiags_amd/lib/libhrfd.so; else L=hackrfdiags_amd/lib/variants/$v/libhrfd.so; fi
    echo -n "$v: "; HRFD_LIB=$PWD/$L python3 tools/mod_time.py $W || exit 1
  done
done
