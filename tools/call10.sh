cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c10; O=$GRAFT_REPO_ROOT/gpurun_out/c10; export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_h2 -o r -- python $GRAFT_REPO_ROOT/tools/halo_h2_bench.py > $O/log.txt 2>&1
python3 - <<PY > $O/sq_halo_h2.txt
import csv, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open('/tmp/pmc_h2/r_counter_collection.csv')):
    k=r['Kernel_Name'][:80]
    if 'halo' not in k: continue
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k,d in agg.items():
    print(k)
    for c,v in sorted(d.items()): print("   %-32s %.5g per dispatch"%(c, v/cnt[(k,c)]))
PY
cat $O/sq_halo_h2.txt
