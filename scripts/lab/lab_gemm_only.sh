#!/bin/bash
# Lab (GPU box): gemm_h.hip rebuilt with extra -D flags and linked with the in-tree objects of the other sources -> /tmp/libofb_lab.so
# usage: OFB_LAB_DEFS="-DOFB_GEMM_H_LAB" bash scripts/lab/lab_gemm_only.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
C=$R/once-for-both_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function ${OFB_LAB_DEFS} -c $C/gemm_h.hip -o /tmp/gemm_h_lab.o 2>/dev/null || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libofb_lab.so /tmp/gemm_h_lab.o $(ls $C/build/*.o | grep -v gemm_h.o) || exit 1
