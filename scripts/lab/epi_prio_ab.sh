#!/bin/bash
# Lab (GPU box): the direct epilogue at raised wave priority (-DOFB_LAB_EPI_PRIO=n) against the product build; the twelve block products.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
C=$R/once-for-both_amd/csrc
mkdir -p /tmp/ofb_prio
pids=""
for a in ${PRIOS:-1 3}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function ${PRIO_DEF:--DOFB_EPI_PRIO}=$a -c $C/gemm_h.hip -o /tmp/ofb_prio/gemm_h_$a.o 2>/dev/null &
  pids="$pids $!"
done
for p in $pids; do wait $p || exit 1; done
others=$(ls $C/build/*.o | grep -v gemm_h.o)
for a in ${PRIOS:-1 3}; do hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/ofb_prio/libofb_p$a.so /tmp/ofb_prio/gemm_h_$a.o $others || exit 1; done
for rep in 1 2; do
  echo "=== product build (run $rep)"; python3 $R/scripts/gemm_step_shapes.py 2>/dev/null | head -13 | grep "${ROWS:-fc1  KC\|dH fc2\|sum over}"
  for a in ${PRIOS:-1 3}; do
    echo "=== direct epilogue at s_setprio $a (run $rep)"
    OFB_LIB_PATH=/tmp/ofb_prio/libofb_p$a.so python3 $R/scripts/gemm_step_shapes.py 2>/dev/null | head -13 | grep "${ROWS:-fc1  KC\|dH fc2\|sum over}"
  done
done
