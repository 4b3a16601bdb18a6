#!/bin/bash
# Lab (GPU box): a stamped build of the GEMM next to the product objects, then the stamp report of the given products.
# usage: bash scripts/lab/stamp_gemm_h.sh qkv fc1 ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
C=$R/once-for-both_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DOFB_H_STAMPS ${OFB_LAB_DEFS} -c $C/gemm_h.hip -o /tmp/gemm_h_stamps.o || exit 1
objs=$(ls $C/build/*.o | grep -v gemm_h.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libofb_stamps.so /tmp/gemm_h_stamps.o $objs || exit 1
for w in "$@"; do OFB_LIB_PATH=/tmp/libofb_stamps.so python3 $R/scripts/lab/stamp_gemm_h.py $w 2>&1 | grep -v amdgpu.ids; done
