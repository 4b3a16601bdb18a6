#!/bin/bash
# PMC passes over the GEMM microbench (diagnostic only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LIB=${1:-$R/once-for-both_amd/csrc/libofb_hip.so}
pass() {
  rm -rf $R/gpurun_out/pmc_gemm
  rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gemm -- python3 $R/scripts/lab/time_gemm.py $LIB main > $R/gpurun_out/pmc_gemm.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob('$R/gpurun_out/pmc_gemm/*/*counter_collection.csv')[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if 'gemm' in r['Kernel_Name']:
        agg[(r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print('grid', k, {c: round(sum(x)/len(x)) for c, x in v.items()})
PY
}
pass "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
pass "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES"
pass "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES SQ_IFETCH SQ_WAIT_IFETCH SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"
