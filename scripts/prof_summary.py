"""per-step kernel summary of a rocprofv3 --kernel-trace --stats run of bench.py: python scripts/prof_summary.py <dir> <steps incl. init/warm-up>"""
import csv, glob, sys
d, steps = sys.argv[1], float(sys.argv[2])
f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'kernel time {tot / 1e6 / steps:.2f} ms/step over {len(rows)} kernels, {sum(int(r["Calls"]) for r in rows) / steps:.0f} launches/step')
fam = {}
for r in rows:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    key = 'gemm_p (all)' if n.startswith('gemm_p_') else ('ATen / runtime' if n.startswith('at::') or n.startswith('__amd') else n.split('(')[0].split('<')[0])
    a = fam.setdefault(key, [0, 0.0])
    a[0] += int(r['Calls']); a[1] += float(r['TotalDurationNs'])
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f'{k[:60]:60s} {c / steps:7.1f}/step {t / 1e6 / steps:8.3f} ms/step  avg {t / c / 1e3:8.1f} us')
