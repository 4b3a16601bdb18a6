#!/bin/bash
# Lab (GPU box): attention backward with the K fragments held across the two query tiles (product, OFB_ATT_KB_HOLD=1) against the
# round-5 read order (=0), alternating runs of scripts/att_perf.py on the same box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
C=$R/once-for-both_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DOFB_ATT_KB_HOLD=0 -c $C/attention.hip -o /tmp/att_kb0.o || exit 1
objs=$(ls $C/build/*.o | grep -v attention.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libofb_kb0.so /tmp/att_kb0.o $objs || exit 1
for i in 1 2 3; do
  python3 $R/scripts/att_perf.py "held (product)" 2>&1 | grep -v amdgpu.ids
  OFB_LIB_PATH=/tmp/libofb_kb0.so python3 $R/scripts/att_perf.py "re-read (round 5)" 2>&1 | grep -v amdgpu.ids
done
