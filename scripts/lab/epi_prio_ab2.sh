#!/bin/bash
# Lab (GPU box): wave priority of the direct epilogue, all three builds made the same way on the box (-DOFB_EPI_PRIO=0 / 1 / 2),
# three interleaved rounds, the two products that take the direct epilogue + the sum of the twelve.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
C=$R/once-for-both_amd/csrc
mkdir -p /tmp/ofb_prio2
pids=""
for a in 0 1 2; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DOFB_EPI_PRIO=$a -c $C/gemm_h.hip -o /tmp/ofb_prio2/gemm_h_$a.o 2>/dev/null &
  pids="$pids $!"
done
for p in $pids; do wait $p || exit 1; done
others=$(ls $C/build/*.o | grep -v gemm_h.o)
for a in 0 1 2; do hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/ofb_prio2/libofb_p$a.so /tmp/ofb_prio2/gemm_h_$a.o $others || exit 1; done
for rep in 1 2 3; do
  for a in 0 1 2; do
    echo "=== OFB_EPI_PRIO=$a (round $rep)"
    OFB_LIB_PATH=/tmp/ofb_prio2/libofb_p$a.so python3 $R/scripts/gemm_step_shapes.py 2>/dev/null | head -13 | grep "fc1  KC\|dH fc2\|sum over"
  done
done
