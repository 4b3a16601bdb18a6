#!/bin/bash
# Lab (GPU box): per-kernel averages of the once-per-step small kernels, previous build (scripts/lab/libofb_prev.so) against the tree's.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for which in prev new; do
  if [ $which = prev ]; then export OFB_LIB_PATH=$R/scripts/lab/libofb_prev.so; else unset OFB_LIB_PATH; fi
  rm -rf /tmp/sk_$which
  OFB_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sk_$which -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof > /tmp/sk_$which.log 2>&1 || exit 1
  echo "== $which"
  python3 - $which <<'PY'
import csv, glob, sys
f = glob.glob(f'/tmp/sk_{sys.argv[1]}/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if any(k in n for k in ('gates_fwd', 'gates_bwd', 'hstat', 'spars_finalize', 'flops_loss', 'to_hformat_kernel', 'to_hformat_multi', 'ln_bwd_p', 'ln_fwd_p', 'colsum_multi', 'embed_assemble', 'patchify', 'norm_targets', 'adamw', 'ln_bwd_kernel', 'ln_bwd_stat')):
        import re
        k = re.search(r'(\w+_kernel)', n)
        print(f"{(k.group(1) if k else n[:40]):34s} calls {r['Calls']:>6s} avg {float(r['AverageNs']) / 1e3:8.1f} us")
PY
done
