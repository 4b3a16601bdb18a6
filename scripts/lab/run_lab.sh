#!/bin/bash
# Ablation lab for the GEMM main loop (pass -D flags; do not use single-letter macro names: they collide with parameter names) (diagnostic only; builds throw-away variants under /tmp on the GPU box).
cd "$(dirname "$0")/../.."
SRC=once-for-both_amd/csrc
i=0
for v in "$@"; do
  i=$((i+1))
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude $v $SRC/gemm.hip $SRC/prof.hip -o /tmp/libgemm_$i.so 2>/dev/null || { echo "build failed: $v"; continue; }
  python scripts/lab/${LAB_SCRIPT:-time_gemm.py} /tmp/libgemm_$i.so "$v"
done
