"""Text summaries of one scripts/round_profiles.sh run:  python scripts/profile_summaries.py gpurun_out/<tag> <tag>
Writes gpurun_out/<tag>/summaries/<tag>_{bench.json, kernel_summary_default_cmd.txt, kernel_summary_sidestream_off.txt,
kernel_stats_*.csv, pmc_traffic_summary.txt, mfma_util_summary.txt}: copy that directory's files into profiles/ and commit them.
Every input is looked up INSIDE the run's own directory (newest file when rocprofv3 wrote several)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

O, tag = sys.argv[1], sys.argv[2]
S = os.path.join(O, 'summaries')
os.makedirs(S, exist_ok=True)


def newest(pattern):
    files = glob.glob(os.path.join(O, pattern), recursive=True)
    if not files:
        raise SystemExit(f'no file matches {pattern} under {O}')
    return max(files, key=os.path.getmtime)


def family(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    if n.startswith('gemm_h_fixup'):
        return 'gemm_h (fix-up)'
    if n.startswith('gemm_h_bound'):
        return 'gemm_h (bound pre-kernel)'
    if n.startswith('gemm_h_kernel'):
        return 'gemm_h (main + tail)'
    if n.startswith('at::') or n.startswith('__amd'):
        return 'ATen / runtime'
    return n.split('(')[0].split('<')[0]


def kernel_summary(stats_csv, steps, out_name, extra=''):
    rows = list(csv.DictReader(open(stats_csv)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    lines = [f'# {tag}: rocprofv3 --kernel-trace --stats, {os.path.relpath(stats_csv, O)} ({steps} steps incl. initialisation / warm-up){extra}',
             f'kernel time {tot / 1e6 / steps:.2f} ms/step over {len(rows)} kernels, {sum(int(r["Calls"]) for r in rows) / steps:.0f} launches/step']
    fam = {}
    for r in rows:
        a = fam.setdefault(family(r['Name']), [0, 0.0])
        a[0] += int(r['Calls']); a[1] += float(r['TotalDurationNs'])
    g = [v for k, v in fam.items() if k.startswith('gemm_h')]
    if g:
        lines.append(f'{"gemm_h (all)":60s} {sum(v[0] for v in g) / steps:7.1f}/step {sum(v[1] for v in g) / 1e6 / steps:8.3f} ms/step')
    for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:32]:
        lines.append(f'{k[:60]:60s} {c / steps:7.1f}/step {t / 1e6 / steps:8.3f} ms/step  avg {t / c / 1e3:8.1f} us')
    open(os.path.join(S, out_name), 'w').write('\n'.join(lines) + '\n')
    return lines


bench = json.loads(open(os.path.join(O, 'bench.json')).read().strip().splitlines()[-1])
shutil.copy(os.path.join(O, 'bench.json'), os.path.join(S, f'{tag}_bench.json'))
steps_default = 6 + bench['warmup'] + bench['steps']
f = newest('stats_default/**/*kernel_stats.csv')
shutil.copy(f, os.path.join(S, f'{tag}_kernel_stats_default_cmd.csv'))
print('\n'.join(kernel_summary(f, steps_default, f'{tag}_kernel_summary_default_cmd.txt', ', the default bench command')[:6]))
f = newest('stats_sideoff/**/*kernel_stats.csv')
shutil.copy(f, os.path.join(S, f'{tag}_kernel_stats_sidestream_off.csv'))
print('\n'.join(kernel_summary(f, 12, f'{tag}_kernel_summary_sidestream_off.txt', ', OFB_SIDE_STREAM=0 (kernels serialised: per-kernel times add up)')[:12]))

# ---- steady state: difference of the 20-step and the 12-step side-off runs, per kernel family -------------------------------
_long = glob.glob(os.path.join(O, 'stats_sideoff_long/**/*kernel_stats.csv'), recursive=True)
f2 = max(_long, key=os.path.getmtime) if _long else None
if f2:
    def fams(path):
        out = {}
        for r in csv.DictReader(open(path)):
            a = out.setdefault(family(r['Name']), [0, 0.0])
            a[0] += int(r['Calls']); a[1] += float(r['TotalDurationNs'])
        return out
    fa, fb, ds = fams(f), fams(f2), 8
    diff = {k: (fb[k][0] - fa.get(k, [0, 0.0])[0], fb[k][1] - fa.get(k, [0, 0.0])[1]) for k in fb}
    lines = [f'# {tag}: STEADY-STATE step = (20-step run - 12-step run) / 8 of rocprofv3 --kernel-trace --stats, OFB_SIDE_STREAM=0: model construction, '
             'optimizer-state allocation (2 fills per parameter) and first-step work cancel out',
             f'kernel time {sum(t for _, t in diff.values()) / 1e6 / ds:.2f} ms/step, {sum(c for c, _ in diff.values()) / ds:.0f} launches/step']
    g = [v for k, v in diff.items() if k.startswith('gemm_h')]
    lines.append(f'{"gemm_h (all)":60s} {sum(v[0] for v in g) / ds:7.1f}/step {sum(v[1] for v in g) / 1e6 / ds:8.3f} ms/step')
    for k, (c, t) in sorted(diff.items(), key=lambda kv: -kv[1][1])[:32]:
        if c > 0:
            lines.append(f'{k[:60]:60s} {c / ds:7.1f}/step {t / 1e6 / ds:8.3f} ms/step  avg {t / c / 1e3:8.1f} us')
    open(os.path.join(S, f'{tag}_kernel_summary_steady_state.txt'), 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:12]))

# ---- HBM-side traffic: FETCH_SIZE x2 (gfx950 reports half of a wide coalesced stream, MI355X_MICROARCH.md) + WRITE_SIZE ----
# STEADY-STATE steps only: patch_mask_kernel runs exactly once per step, at its start; the dispatches between the
# first marker of the LAST `steps_pmc` steps and the end of the run are what is counted (model construction, optimizer-state
# allocation and the first step's one-off conversions fall away, as in the steady-state kernel summary)
steps_pmc = 2
MARKER = 'patch_mask_kernel'


def steady_rows(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Dispatch_Id']))
    marks = sorted({int(r['Dispatch_Id']) for r in rows if MARKER in r['Kernel_Name']})
    if len(marks) < steps_pmc + 1:
        raise SystemExit(f'{path}: {len(marks)} step markers, need more than {steps_pmc}')
    first = marks[-steps_pmc]
    return [r for r in rows if int(r['Dispatch_Id']) >= first]


out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in steady_rows(newest(f'pmc_{c}/**/*counter_collection.csv')):
        k = family(r['Kernel_Name'])
        agg[k][0] += 1; agg[k][1] += float(r['Counter_Value'])
    out[c] = agg
lines = [f'# {tag}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, side stream off, the last {steps_pmc} steps of the run = steady state); counters in KiB; FETCH_SIZE doubled (gfx950)']
for k in sorted(out['FETCH_SIZE'], key=lambda k: -out['FETCH_SIZE'][k][1])[:14]:
    n, fs = out['FETCH_SIZE'][k]
    ws = out['WRITE_SIZE'].get(k, [0, 0.0])[1]
    lines.append(f'{k[:44]:44s} launches/step {n / steps_pmc:7.1f}  read {2 * fs * 1024 / steps_pmc / 1e6:9.1f} MB/step  write {ws * 1024 / steps_pmc / 1e6:9.1f} MB/step')
gb = sum(2 * out['FETCH_SIZE'][k][1] + out['WRITE_SIZE'].get(k, [0, 0.0])[1] for k in out['FETCH_SIZE'] if k.startswith('gemm_h')) * 1024 / steps_pmc
lines.append(f'GEMM_BYTES_PER_STEP {gb:.0f}   # all GEMM main / tail / fix-up launches of one step: FETCH_SIZE x2 + WRITE_SIZE (bench.py divides by its GEMM calls per step)')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as _bench
lines.append(f'CSRC_SHA {_bench.csrc_hash()}   # sha256 prefix of csrc/*.hip, csrc/*.h, include/*.h of the profiled build (bench.py quotes this file only for the same build)')
open(os.path.join(S, f'{tag}_pmc_traffic_summary.txt'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines[-3:]))

# ---- matrix-pipe utilisation: SQ_VALU_MFMA_BUSY_CYCLES per kernel family ----
dur = {}
for r in csv.DictReader(open(newest('pmc_sq/**/*kernel_trace.csv'))):
    dur[r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt, seen = collections.Counter(), set()
for r in steady_rows(newest('pmc_sq/**/*counter_collection.csv')):
    k = family(r['Kernel_Name'])
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in seen:
        seen.add(r['Dispatch_Id']); agg[k]['ns'] += dur.get(r['Dispatch_Id'], 0); cnt[k] += 1
fam_t = {}
for r in csv.DictReader(open(newest('stats_sideoff/**/*kernel_stats.csv'))):
    fam_t.setdefault(family(r['Name']), [0, 0.0])
    fam_t[family(r['Name'])][0] += int(r['Calls']); fam_t[family(r['Name'])][1] += float(r['TotalDurationNs'])
lines = [f'# {tag}: matrix-pipe utilisation and memory rate per kernel family (rocprofv3 --pmc, own pass, side stream off, the last {steps_pmc} steps = steady state).',
         '# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES (the SQ block\'s own ratio); pipe share = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x',
         '# launch duration x 2.4 GHz): the fraction of the NOMINAL matrix-pipe cycles that carried an MFMA (comparable with roofline.frac of bench.py);',
         '# clock = GRBM_GUI_ACTIVE / 8 XCDs / duration (reads high on dispatches shorter than ~0.3 ms, MI355X_MICROARCH.md DVFS); HBM GB/s = (FETCH_SIZE x2',
         '# + WRITE_SIZE) / duration of the family in the serialised kernel stats, against the 8000 GB/s spec.']
rate = {}
for k in out['FETCH_SIZE']:
    byts = (2 * out['FETCH_SIZE'][k][1] + out['WRITE_SIZE'].get(k, [0, 0.0])[1]) * 1024 / steps_pmc
    if k in fam_t and fam_t[k][1] > 0:
        rate[k] = byts / (fam_t[k][1] / 12)           # bytes per step / ns per step = GB/s
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]['ns'])[:14]:
    ns = a['ns']
    if not ns:
        continue
    share = a['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * ns * 2.4)
    lines.append(f'{k[:40]:40s} n/step {cnt[k] / steps_pmc:6.1f} avg {ns / cnt[k] / 1e3:7.1f} us  MFMA busy {a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["SQ_BUSY_CYCLES"] + 1e-9):6.3f}'
                 f'  pipe share {share:6.3f}  clock {a["GRBM_GUI_ACTIVE"] / 8 / ns:5.2f} GHz  wait_inst/wave {a["SQ_WAIT_INST_ANY"] / (a["SQ_WAVE_CYCLES"] + 1e-9):5.2f}'
                 f'  wait_any/wave {a["SQ_WAIT_ANY"] / (a["SQ_WAVE_CYCLES"] + 1e-9):5.2f}  HBM {rate.get(k, 0):7.0f} GB/s ({rate.get(k, 0) / 8000:5.1%} of 8 TB/s)')
lines.append(f'# bench line of the same build: {bench["value"]} images/s, {bench["ms_per_step"]} ms/step, GEMM roofline {bench["roofline"]["achieved"] if bench.get("roofline") else None} TFLOP/s')
open(os.path.join(S, f'{tag}_mfma_util_summary.txt'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines[5:]))
print('summaries in', S)
