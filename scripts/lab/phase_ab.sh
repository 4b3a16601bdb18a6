#!/bin/bash
# Lab (GPU box): de-phasing the store bursts of multi-round launches - the workgroups of every other CU start n x 4096 cycles late
# (-DOFB_LAB_PHASE=n) - against the product build (0 is built the same way on the box); qkv / fc1 / dH and the sum of the twelve products.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
C=$R/once-for-both_amd/csrc
mkdir -p /tmp/ofb_ph
pids=""
for a in ${PHASES:-0 2 3 4 6}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DOFB_LAB_PHASE=$a -c $C/gemm_h.hip -o /tmp/ofb_ph/gemm_h_$a.o 2>/dev/null &
  pids="$pids $!"
done
for p in $pids; do wait $p || exit 1; done
others=$(ls $C/build/*.o | grep -v gemm_h.o)
for a in ${PHASES:-0 2 3 4 6}; do hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/ofb_ph/libofb_$a.so /tmp/ofb_ph/gemm_h_$a.o $others || exit 1; done
for rep in 1 2; do
  for a in ${PHASES:-0 2 3 4 6}; do
    echo "=== OFB_LAB_PHASE=$a (round $rep)"
    OFB_LIB_PATH=/tmp/ofb_ph/libofb_$a.so python3 $R/scripts/gemm_step_shapes.py 2>/dev/null | head -13 | grep "qkv  KC\|fc1  KC\|dH fc2\|sum over"
  done
done
