#!/bin/bash
# Round deliverables on one GPU box, every output under a ROUND-UNIQUE directory gpurun_out/<tag>/ (so a stale file of an earlier
# round can never be picked up): (1) the default bench line (with cpu_baseline), (2) rocprofv3 --kernel-trace --stats of that same
# command, (3) the serialised (side stream off) kernel stats, (4) PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes),
# (5) SQ counters (matrix-pipe busy cycles, wave cycles, clock).  scripts/profile_summaries.py turns them into the text summaries
# that are committed under profiles/.          usage (GPU box): bash scripts/round_profiles.sh r03_v1 [quick]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:?usage: round_profiles.sh <round tag, e.g. r03_v1> [quick]}
O=$R/gpurun_out/$tag
rm -rf $O; mkdir -p $O
cd $R
if [ "$2" != "quick" ]; then python3 bench.py > $O/bench.json 2> $O/bench.log || exit 1; else python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.log || exit 1; fi
echo "bench done: $(cut -c1-160 $O/bench.json)"
cd /tmp && export TMPDIR=/tmp
# the program itself goes directly after `--` (no env / bash -c hop under rocprofv3)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -- python3 $R/bench.py --no-cpu-baseline > $O/stats_default.log 2>&1 || exit 1
echo "stats (default command) done"
export OFB_SIDE_STREAM=0      # one stream: per-kernel durations / counters are attributable
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sideoff -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-prof > $O/stats_sideoff.log 2>&1 || exit 1
echo "stats (side stream off) done"
# the same with 8 more timed steps: the difference of the two runs is the STEADY-STATE step (model construction, optimizer-state
# allocation and the first-step work cancel out)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sideoff_long -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-prof > $O/stats_sideoff_long.log 2>&1 || exit 1
echo "stats (side stream off, 8 more steps) done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $O/pmc_$c.log 2>&1 || exit 1
  echo "pmc $c done"
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $O/pmc_sq.log 2>&1 || exit 1
echo "pmc SQ done"
unset OFB_SIDE_STREAM
cd $R && python3 scripts/profile_summaries.py $O $tag
