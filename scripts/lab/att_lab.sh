#!/bin/bash
cd "$(dirname "$0")/../.."
SRC=once-for-both_amd/csrc
i=0
for v in "-DLAB_BASE" "-DLAB_ATT_NOATOMIC" "-DLAB_ATT_NOSTAGE" "-DLAB_ATT_NOATOMIC -DLAB_ATT_NOSTAGE"; do
  i=$((i+1))
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v $SRC/attention.hip $SRC/prof.hip -o /tmp/libatt_$i.so 2>/dev/null || { echo "build failed $v"; continue; }
  python scripts/lab/time_att.py /tmp/libatt_$i.so "$v"
done
