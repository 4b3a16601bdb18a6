#!/bin/bash
# Lab (GPU box): direct epilogue with paired 16-byte plane stores (-DOFB_DIRECT_PAIR=1) against two 8-byte stores per lane (=0); both built on the box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
C=$R/once-for-both_amd/csrc
mkdir -p /tmp/ofb_pair
pids=""
for a in 0 1; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DOFB_DIRECT_PAIR=$a -c $C/gemm_h.hip -o /tmp/ofb_pair/gemm_h_$a.o 2>/dev/null &
  pids="$pids $!"
done
for p in $pids; do wait $p || exit 1; done
others=$(ls $C/build/*.o | grep -v gemm_h.o)
for a in 0 1; do hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/ofb_pair/libofb_$a.so /tmp/ofb_pair/gemm_h_$a.o $others || exit 1; done
for rep in 1 2 3; do
  for a in 0 1; do
    echo "=== OFB_DIRECT_PAIR=$a (round $rep)"
    OFB_LIB_PATH=/tmp/ofb_pair/libofb_$a.so python3 $R/scripts/gemm_step_shapes.py 2>/dev/null | head -13 | grep "fc1  KC\|dH fc2\|sum over"
  done
done
