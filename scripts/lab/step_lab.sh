#!/bin/bash
# build the whole library with extra -D flags into /tmp and time GEMM shapes against it (diagnostic)
cd "$(dirname "$0")/../.."
SRC=once-for-both_amd/csrc
SCRIPT=${SCRIPT:-scripts/gemm_step_shapes.py}
for v in "$@"; do
  rm -rf /tmp/labrepo && mkdir -p /tmp/labrepo && cp -r once-for-both_amd ofb_amd.py scripts include /tmp/labrepo/
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v $SRC/*.hip -o /tmp/labrepo/once-for-both_amd/csrc/libofb_hip.so 2>/dev/null || { echo "build failed: $v"; continue; }
  echo "=== $v"
  (cd /tmp/labrepo && python $SCRIPT 2>&1 | grep -v amdgpu)
done
