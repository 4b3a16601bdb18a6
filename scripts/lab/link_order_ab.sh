#!/bin/bash
# Lab (GPU box): does the position of the GEMM's code inside libofb_hip.so matter?  The product library (built in the container, objects
# linked in source-name order) against libraries linked on the box from the SAME in-tree objects: same order, gemm_h.o first, gemm_h.o last.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
C=$R/once-for-both_amd/csrc
mkdir -p /tmp/ofb_lo
all=$(ls $C/build/*.o)
others=$(ls $C/build/*.o | grep -v gemm_h.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/ofb_lo/same.so $all || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/ofb_lo/first.so $C/build/gemm_h.o $others || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/ofb_lo/last.so $others $C/build/gemm_h.o || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -c $C/gemm_h.hip -o /tmp/ofb_lo/gemm_h_box.o 2>/dev/null || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/ofb_lo/boxobj_first.so /tmp/ofb_lo/gemm_h_box.o $others || exit 1
cmp $C/build/gemm_h.o /tmp/ofb_lo/gemm_h_box.o && echo "gemm_h.o: container build == box build (byte-identical)" || echo "gemm_h.o: container build != box build"
hipcc --version | head -2
for rep in 1 2; do
  for v in product same first last boxobj_first; do
    echo "=== $v (round $rep)"
    if [ $v = product ]; then python3 $R/scripts/gemm_step_shapes.py 2>/dev/null | head -13 | grep "${ROWS:-qkv  KC\|fc1  KC\|dH fc2\|sum over}"
    else OFB_LIB_PATH=/tmp/ofb_lo/$v.so python3 $R/scripts/gemm_step_shapes.py 2>/dev/null | head -13 | grep "${ROWS:-qkv  KC\|fc1  KC\|dH fc2\|sum over}"; fi
  done
done
