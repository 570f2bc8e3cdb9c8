#!/bin/bash
# Register / LDS / scratch use of the hrfd kernels: compiles hrfd_lib.hip with --save-temps into
# /tmp/dis and prints the kernel metadata (the .s file stays there for reading the ISA).
set -e
cd "$(dirname "$0")/../hackrfdiags_amd/csrc"
mkdir -p /tmp/dis
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function \
  --save-temps=obj $EXTRA -c -o /tmp/dis/hrfd_lib.o hrfd_lib.hip 2>/dev/null
S=/tmp/dis/hrfd_lib-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "^    \.name:|\.vgpr_count|\.sgpr_count|group_segment_fixed_size|private_segment_fixed_size|vgpr_spill_count" $S | paste - - - - - - | sed 's/  */ /g' | grep -E "${1:-.}"
