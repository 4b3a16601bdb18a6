#!/bin/bash
# Lab (GPU box): attention forward with the staging done by the nine waves off SIMD 0 (product, OFB_ATT_STAGE_OFF_SIMD0=1) against all thirteen
# waves staging (=0: rounds 3-6), alternating runs of scripts/att_perf.py on the same box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
C=$R/once-for-both_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DOFB_ATT_STAGE_OFF_SIMD0=0 -c $C/attention.hip -o /tmp/att_st0.o || exit 1
objs=$(ls $C/build/*.o | grep -v attention.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libofb_st0.so /tmp/att_st0.o $objs || exit 1
for i in 1 2 3; do
  python3 $R/scripts/att_perf.py "nine stagers (off SIMD 0)" 2>&1 | grep -v amdgpu.ids
  OFB_LIB_PATH=/tmp/libofb_st0.so python3 $R/scripts/att_perf.py "thirteen stagers" 2>&1 | grep -v amdgpu.ids
done
