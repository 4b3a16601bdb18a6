#!/bin/bash
# Same-box A/B of gemm_p.hip build variants: builds libofb_hip.<tag>.so per -D flag set HERE (hipcc cross-compiles), the GPU side
# runs scripts/gemm_step_shapes_p.py (or bench.py with BENCH=1) against each through OFB_LIB_PATH.
# usage (container): scripts/lab/ab_gemm_p.sh build "tagA:<flags>" "tagB:<flags>" ;  (GPU box) scripts/lab/ab_gemm_p.sh run tagA tagB
cd "$(dirname "$0")/../.."
SRC=once-for-both_amd/csrc
V=$SRC/build/variants
mkdir -p $V
mode=$1; shift
if [ "$mode" = build ]; then
  OBJS=$(ls $SRC/build/*.o | grep -v "/gemm_p.o")
  for spec in "$@"; do
    tag=${spec%%:*}; flags=${spec#*:}
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude $flags -c $SRC/gemm_p.hip -o $V/gemm_p.$tag.o || exit 1
    hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libofb_hip.$tag.so $V/gemm_p.$tag.o $OBJS || exit 1
    echo "built $tag ($flags)"
  done
else
  for rep in 1 2; do
    for tag in "$@"; do
      echo "== $tag (rep $rep)"
      if [ -n "$BENCH" ]; then
        OFB_LIB_PATH=$PWD/$V/libofb_hip.$tag.so python bench.py --no-cpu-baseline --no-prof --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
      else
        OFB_LIB_PATH=$PWD/$V/libofb_hip.$tag.so python scripts/gemm_step_shapes_p.py | grep -v convert
      fi
    done
  done
fi
