#!/bin/bash
# Register / LDS / scratch use of the hrfd kernels: compiles hrfd_lib.hip with --save-temps into
# /tmp/dis and prints the kernel metadata (the .s file stays there for reading the ISA).
# usage: [EXTRA="-D..."] tools/kinfo.sh [name regex]
set -e
cd "$(dirname "$0")/../hackrfdiags_amd/csrc"
mkdir -p /tmp/dis
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function \
  --save-temps=obj $EXTRA -c -o /tmp/dis/hrfd_lib.o hrfd_lib.hip 2>/dev/null
python3 - "${1:-.}" <<'PY'
import re, sys
pat = re.compile(sys.argv[1])
txt = open('/tmp/dis/hrfd_lib-hip-amdgcn-amd-amdhsa-gfx950.s').read()
meta = txt[txt.index('amdhsa.kernels:'):]
for blk in re.split(r'\n  - ', meta)[1:]:
    d = dict(re.findall(r'\.(name|vgpr_count|sgpr_count|vgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):\s+(\S+)', blk))
    if 'name' in d and pat.search(d['name']):
        print(f"{d['name'][:60]:60s} vgpr {d.get('vgpr_count'):>4s} spill {d.get('vgpr_spill_count'):>3s} sgpr {d.get('sgpr_count'):>4s} lds {d.get('group_segment_fixed_size'):>7s} scratch {d.get('private_segment_fixed_size'):>5s}")
PY
