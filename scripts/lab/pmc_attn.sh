#!/bin/bash
# SQ counters of the attention kernels in the real step (two passes of <= 8 counters)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export OFB_SIDE_STREAM=0
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmc_att$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_att$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-prof > $R/gpurun_out/pmc_att$i.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections
for i in (1, 2):
    f = glob.glob('$R/gpurun_out/pmc_att%d/*/*counter_collection.csv' % i)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'attn_' not in n: continue
        k = 'attn_fwd' if 'fwd' in n else 'attn_bwd'
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
    for k, a in agg.items():
        n = len(cnt[k])
        print(k, 'launches', n, {c: round(v / n) for c, v in a.items()})
PY
