#!/bin/bash
# lab build of the library with in-kernel stamps in the attention backward (scripts/lab/stamp_att.py)
set -e
cd "$(dirname "$0")/../.."
CS=once-for-both_amd/csrc
mkdir -p scripts/lab/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DOFB_ATT_STAMPS -c $CS/attention.hip -o scripts/lab/bin/attention_stamps.o
OBJS=$(ls $CS/build/*.o | grep -v attention.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/lab/bin/libofb_attstamps.so scripts/lab/bin/attention_stamps.o $OBJS
