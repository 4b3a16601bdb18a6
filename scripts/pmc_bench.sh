#!/bin/bash
# HBM traffic of the bench's kernels: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) as MI355X_MICROARCH.md prescribes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export OFB_SIDE_STREAM=0      # one stream: per-kernel counters are attributable
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $R/gpurun_out/pmc_$c.log 2>&1
done
python3 - <<PY
import csv, glob, collections
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('$R/gpurun_out/pmc_%s/*/*counter_collection.csv' % c)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        key = ('gemm_fixup_kernel' if 'fixup' in n else 'gemm_kernel') if ('gemm_p_' in n or 'gemm_f32' in n or 'gemm_fixup' in n) else ('to_pformat (conversions)' if 'to_pformat' in n else ('attn_bwd_kernel' if 'attn_bwd' in n else ('attn_fwd_kernel' if 'attn_fwd' in n else ('ln_kernels' if 'ln_' in n else ('colsum kernels' if 'colsum' in n else n.split('(')[0][-40:])))))
        agg[key][0] += 1; agg[key][1] += float(r['Counter_Value'])
    out[c] = agg
steps = 9          # bench.py: 6 initialisation steps + --warmup 1 + --steps 2
lines = []
for k in sorted(out['FETCH_SIZE'], key=lambda k: -out['FETCH_SIZE'][k][1])[:12]:
    n, fs = out['FETCH_SIZE'][k]; ws = out['WRITE_SIZE'].get(k, [0, 0.0])[1]
    # counters are in KiB; gfx950 FETCH_SIZE reports half of a wide coalesced stream -> doubled (MI355X_MICROARCH.md, HBM)
    lines.append(f'{k:42s} launches/step {n/steps:7.1f}  read {2*fs*1024/steps/1e6:9.1f} MB/step (FETCH_SIZE x2)  write {ws*1024/steps/1e6:9.1f} MB/step')
print('\n'.join(lines))
g = lambda c, k: out[c].get(k, [0, 0.0])[1] * 1024 / steps
gemm_bytes = 2 * (g('FETCH_SIZE', 'gemm_kernel') + g('FETCH_SIZE', 'gemm_fixup_kernel')) + g('WRITE_SIZE', 'gemm_kernel') + g('WRITE_SIZE', 'gemm_fixup_kernel')
lines.append(f'GEMM_BYTES_PER_STEP {gemm_bytes:.0f}   # all GEMM main / tail / fix-up launches of one step: FETCH_SIZE x2 + WRITE_SIZE (bench.py divides by its 152 GEMM calls)')
print(lines[-1])
open('$R/gpurun_out/pmc_traffic_summary.txt', 'w').write('\n'.join(lines) + '\n')
PY
