#!/bin/bash
# L2 hit rate / fabric traffic of the GEMM main loops (diagnostic): product kernel (time_gemm2.py forms) and the lab kernels.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
summ() {
python3 - "$1" <<PY
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'gemm' not in n or 'fixup' in n or 'split_image' in n: continue
    key = n.split('(')[0][-60:] + ' grid' + r['Grid_Size']
    agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k, {c: round(sum(x)/len(x)) for c, x in v.items()}, 'n', len(next(iter(v.values()))))
PY
}
for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rm -rf $R/gpurun_out/pmc_l2_$tag
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_l2_$tag -- $R/scripts/lab/bin/gemm_r_lab_s2w2 > /dev/null 2>&1
  echo "== lab R: $c"; summ $R/gpurun_out/pmc_l2_$tag
  rm -rf $R/gpurun_out/pmc_l2p_$tag
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_l2p_$tag -- python3 $R/scripts/lab/time_gemm2.py $R/once-for-both_amd/csrc/libofb_hip.so main > /dev/null 2>&1
  echo "== product: $c"; summ $R/gpurun_out/pmc_l2p_$tag
done
