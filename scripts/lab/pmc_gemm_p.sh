#!/bin/bash
# matrix-pipe utilisation and held clock of the P-engine GEMM kernels in the real step (SQ counters, one pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export OFB_SIDE_STREAM=0
rm -rf $R/gpurun_out/pmc_sq
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $R/gpurun_out/pmc_sq.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob('$R/gpurun_out/pmc_sq/*/*counter_collection.csv')[0]
tr = glob.glob('$R/gpurun_out/pmc_sq/*/*kernel_trace.csv')[0]
dur = {}
for r in csv.DictReader(open(tr)):
    dur[r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '')
    if 'gemm_p_kernel' not in n and 'attn_' not in n: continue
    key = n.split('(')[0][-48:]
    agg[key][r['Counter_Name']] += float(r['Counter_Value'])
    if (r['Dispatch_Id']) not in seen:
        seen.add(r['Dispatch_Id']); agg[key]['ns'] += dur.get(r['Dispatch_Id'], 0); cnt[key] += 1
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]['ns'])[:12]:
    ns = a['ns']
    # SQ_VALU_MFMA_BUSY_CYCLES: cycles summed over SIMDs?  report ratios that are unit-safe
    clk = a['GRBM_GUI_ACTIVE'] / 8 / ns if ns else 0            # GHz (sum over 8 XCDs / 8 / ns)
    mfma = a['SQ_VALU_MFMA_BUSY_CYCLES'] / (a['SQ_BUSY_CYCLES'] + 1e-9)
    print(f'{k:50s} n {cnt[k]:4d} avg {ns/cnt[k]/1e3:7.1f} us  clock {clk:5.2f} GHz  MFMA_BUSY/SQ_BUSY {mfma:6.3f}  WAIT_INST/WAVE {a["SQ_WAIT_INST_ANY"]/(a["SQ_WAVE_CYCLES"]+1e-9):5.2f} WAIT_ANY/WAVE {a["SQ_WAIT_ANY"]/(a["SQ_WAVE_CYCLES"]+1e-9):5.2f} ACTIVE/WAVE {a["SQ_ACTIVE_INST_ANY"]/(a["SQ_WAVE_CYCLES"]+1e-9):5.2f}')
PY
