#!/bin/bash
# SQ counters for the conv kernels at the benchmark layer shapes (own run, kernel-trace only)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc1 -o p -- python $R/tools/conv_bench.py fp16 > /tmp/pmc1.log 2>&1
echo rc=$?; tail -3 /tmp/pmc1.log | cut -c1-200
python3 - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('/tmp/pmc1/p_counter_collection.csv')))
agg=collections.OrderedDict()
for r in rows:
    n=r['Kernel_Name']
    if 'igemm' not in n and 'wgrad' not in n: continue
    key=(n[:70], r['Grid_Size'])
    d=agg.setdefault(key, collections.defaultdict(float))
    d[r['Counter_Name']]+=float(r['Counter_Value']); d['_n']+=1
with open('/root/repo/gpurun_out/pmc/sq_counters.txt','w') as f:
    for (n,g),d in agg.items():
        c=d['_n']/8
        wc=d['SQ_WAVE_CYCLES'] or 1
        line="%-72s grid=%9s calls=%3d  wait_any/wave=%.2f wait_inst/wave=%.2f active/wave=%.2f  mfma_busy/busy=%.3f  lds_conflict/lds_active=%.3f"%(n,g,c,d['SQ_WAIT_ANY']/wc,d['SQ_WAIT_INST_ANY']/wc,d['SQ_ACTIVE_INST_ANY']/wc,d['SQ_VALU_MFMA_BUSY_CYCLES']/(d['SQ_BUSY_CYCLES'] or 1),d['SQ_LDS_BANK_CONFLICT']/(d['SQ_LDS_IDX_ACTIVE'] or 1))
        print(line); f.write(line+"\n")
PY
