#!/bin/bash
# Lab (GPU box): the whole library rebuilt with extra -D flags (OFB_LAB_DEFS) -> /tmp/libofb_lab.so; run anything against it with
# OFB_LIB_PATH=/tmp/libofb_lab.so.    usage: OFB_LAB_DEFS="-DOFB_LAB_NO_NT" bash scripts/lab/lab_build_all.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
C=$R/once-for-both_amd/csrc
mkdir -p /tmp/ofb_lab_obj
pids=""
for f in $C/*.hip; do
  b=$(basename $f .hip)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I$R/include ${OFB_LAB_DEFS} -c $f -o /tmp/ofb_lab_obj/$b.o &
  pids="$pids $!"
done
for f in $C/*.cpp; do
  b=$(basename $f .cpp)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -I$R/include ${OFB_LAB_DEFS} -c $f -o /tmp/ofb_lab_obj/$b.o &
  pids="$pids $!"
done
for p in $pids; do wait $p || exit 1; done
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libofb_lab.so /tmp/ofb_lab_obj/*.o || exit 1
