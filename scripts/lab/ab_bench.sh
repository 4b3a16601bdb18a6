#!/bin/bash
# A/B of whole-step throughput for gemm.hip build variants ON ONE BOX (boxes differ by ~5 %): rebuilds the in-tree library with
# each -D flag set in turn, runs bench.py, restores the default build at the end.  usage: ab_bench.sh "<flags A>" "<flags B>" ...
cd "$(dirname "$0")/../.."
SRC=once-for-both_amd/csrc
OBJS=$(ls $SRC/build/*.o | grep -v "/gemm.o")
cp $SRC/libofb_hip.so /tmp/libofb_hip.orig.so
for rep in 1 2; do
for v in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude $v -c $SRC/gemm.hip -o /tmp/gemm_ab.o 2>/dev/null || { echo "build failed: $v"; continue; }
  hipcc --offload-arch=gfx950 -shared -o $SRC/libofb_hip.so /tmp/gemm_ab.o $OBJS 2>/dev/null || { echo "link failed: $v"; continue; }
  r=$(python bench.py --no-cpu-baseline --no-prof --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "$v : $r"
done
done
cp /tmp/libofb_hip.orig.so $SRC/libofb_hip.so
