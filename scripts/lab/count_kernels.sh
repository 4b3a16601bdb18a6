#!/bin/bash
# launches per step by kernel at a launch-bound size: scripts/lab/count_kernels.sh [model] [batch]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; m=${1:-deit_tiny}; b=${2:-8}
rm -rf $R/gpurun_out/cnt
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cnt -- python3 $R/bench.py --model $m --batch $b --steps 4 --warmup 2 --no-cpu-baseline --no-prof > $R/gpurun_out/cnt.log 2>&1 || exit 1
python3 - <<PY
import csv, glob
f = glob.glob('$R/gpurun_out/cnt/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 12.0
tot = sum(int(r['Calls']) for r in rows) / steps
print(f'{tot:.0f} launches per step, {sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6:.2f} ms of kernels per step')
for r in sorted(rows, key=lambda r: -int(r['Calls']))[:40]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print(f"{int(r['Calls']) / steps:7.1f}/step  avg {float(r['AverageNs']) / 1e3:6.1f} us  {n[:120]}")
PY
