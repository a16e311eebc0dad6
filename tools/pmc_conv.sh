#!/bin/bash
# SQ counters of the conv kernels on one layer shape: tools/pmc_conv.sh "256->256" -> gpurun_out/pmc_conv_<tag>.txt
cd "$GRAFT_REPO_ROOT" || exit 1
M="${1:-256->256}"; TAG="${2:-l3}"; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL"; do
  n=$(echo $set | cut -d' ' -f1)
  (cd /tmp && CB_MATCH="$M" timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$n -o r -- python $R/tools/conv_bench.py > /dev/null 2>&1)
  python3 - <<PY >> $R/gpurun_out/pmc_conv_$TAG.txt
import csv, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
try:
    for r in csv.DictReader(open('/tmp/pmc_$n/r_counter_collection.csv')):
        k=r['Kernel_Name'][:70]
        if 'igemm' not in k and 'wgrad' not in k and 'halo' not in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
    for k,d in agg.items():
        print(k)
        for c,v in d.items(): print("   %-32s %.4g per dispatch"%(c, v/cnt[(k,c)]))
except Exception as e:
    print("failed", e)
PY
done
