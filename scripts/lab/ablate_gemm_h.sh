#!/bin/bash
# Lab (GPU box): what is the time of the 16x16x32 GEMM kernel made of?  Builds gemm_h.hip five times with ONE ingredient removed each
# (-DOFB_LAB_ABLATE=n, results are wrong by construction; ABL="6 7 8 9" selects the builds), links each with the in-tree objects of the other sources and times the twelve
# block products of a DeiT-S step on every build.    usage: bash scripts/lab/ablate_gemm_h.sh > gpurun_out/ablate.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
C=$R/once-for-both_amd/csrc
mkdir -p /tmp/ofb_abl
pids=""
for a in ${ABL:-1 2 3 4 5}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DOFB_LAB_ABLATE=$a -c $C/gemm_h.hip -o /tmp/ofb_abl/gemm_h_$a.o 2>/dev/null &
  pids="$pids $!"
done
for p in $pids; do wait $p || exit 1; done
others=$(ls $C/build/*.o | grep -v gemm_h.o)
for a in ${ABL:-1 2 3 4 5}; do
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/ofb_abl/libofb_a$a.so /tmp/ofb_abl/gemm_h_$a.o $others || exit 1
done
echo "=== full kernel"; python3 $R/scripts/gemm_step_shapes.py 2>/dev/null | head -13
names=("" "no B pieces after the prologue" "no LDS-DMA after the prologue" "no MFMAs" "no fragment reads after the prologue" "no global stores in the wide epilogue" "no GELU-derivative (aux) stores" "no plane stores" "GELU pieces replaced by two FMAs" "no epilogue at all")
for a in ${ABL:-1 2 3 4 5}; do
  echo "=== ablation $a: ${names[$a]}"
  OFB_LIB_PATH=/tmp/ofb_abl/libofb_a$a.so python3 $R/scripts/gemm_step_shapes.py 2>/dev/null | head -13
done
