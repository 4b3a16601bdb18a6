#!/bin/bash
# rocprofv3 kernel stats of the default bench step with the side stream off (serialised kernels: per-kernel times add up), then the
# un-profiled bench line.  usage (GPU box): scripts/prof_bench.sh <tag>   -> gpurun_out/prof_<tag>/, gpurun_out/prof_<tag>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-run}
rm -rf $R/gpurun_out/prof_$tag
OFB_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-prof > $R/gpurun_out/prof_$tag.log 2>&1 || exit 1
# 12 = 6 initialisation + 2 warm-up + 4 timed steps
python3 $R/scripts/prof_summary.py $R/gpurun_out/prof_$tag 12 30 > $R/gpurun_out/prof_$tag.txt
cd $R && python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof 2>/dev/null | tail -1 >> $R/gpurun_out/prof_$tag.txt
cat $R/gpurun_out/prof_$tag.txt
