#!/bin/bash
# Round deliverables on one GPU box: (1) the default bench line (with cpu_baseline), (2) rocprofv3 --kernel-trace --stats of that
# same command, (3) the serialised (side stream off) kernel summary, (4) PMC traffic.  usage: scripts/round_profiles.sh <tag>
R=$GRAFT_REPO_ROOT; tag=${1:-v2}
cd $R && python3 bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.log || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/stats_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_$tag -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/stats_$tag.log 2>&1 || exit 1
cp $(ls $R/gpurun_out/stats_$tag/*/*kernel_stats.csv | head -1) $R/gpurun_out/kernel_stats_default_$tag.csv
cd $R && scripts/prof_bench.sh $tag > /dev/null 2>&1 || exit 1
bash scripts/pmc_bench.sh > gpurun_out/pmc_$tag.log 2>&1 || exit 1
tail -3 gpurun_out/pmc_$tag.log; tail -1 gpurun_out/bench_$tag.json | cut -c1-400
