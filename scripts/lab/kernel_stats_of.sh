#!/bin/bash
# Lab (GPU box): rocprofv3 --kernel-trace --stats of one bench.py invocation (side stream off), top kernels by total time.
# usage: bash scripts/lab/kernel_stats_of.sh --mode finetune --batch 256
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kso
OFB_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kso -- python3 $R/bench.py "$@" --steps 10 --warmup 3 --no-cpu-baseline --no-prof > /tmp/kso.log 2>&1 || { tail -n 5 /tmp/kso.log; exit 1; }
python3 - <<'PY'
import csv, glob, re
f = glob.glob('/tmp/kso/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'total kernel time {tot / 1e6:.1f} ms over the run (13 steps + set-up)')
for r in rows[:30]:
    m = re.search(r'(\w+_kernel)', r['Name']); k = m.group(1) if m else r['Name'][:44]
    print(f"{k[:44]:44s} calls {r['Calls']:>6s} total {float(r['TotalDurationNs']) / 1e6:8.2f} ms avg {float(r['AverageNs']) / 1e3:8.1f} us {100 * float(r['TotalDurationNs']) / tot:5.1f}%")
PY
