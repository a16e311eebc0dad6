#!/bin/bash
# round 3, GPU call 16: BatchNorm-backward sums taken in the data-gradient epilogues (MN_FUSE_BN_SUMS=0 restores the reduction launches)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c16; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "batchnorm_sums or chunk_resident or full_size or train_step or trajectory" 2>&1 | tail -3 | tee $O/pytest.txt
for rep in 1 2 3; do for f in 0 1; do
  MN_FUSE_BN_SUMS=$f timeout 300 python bench.py --steps 50 --repeats 3 --no-cpu-baseline --no-parity-mode --no-events > $O/bench.json 2>> $O/bench.err
  python3 -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('fuse $f fp16', d['value'], d['ms_per_step'], d['config']['region_ms_per_step'])" | tee -a $O/bench_fuse.txt
done; done
for f in 0 1; do
  MN_FUSE_BN_SUMS=$f timeout 300 python bench.py --dtype fp32x3 --steps 20 --repeats 3 --no-cpu-baseline --no-parity-mode --no-events > $O/bench.json 2>> $O/bench.err
  python3 -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('fuse $f fp32x3', d['value'], d['ms_per_step'], d['config']['region_ms_per_step'])" | tee -a $O/bench_fuse.txt
done
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && MN_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_serial -o r -- python $R/bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-events --no-parity-mode > $R/$O/rocprof.log 2>&1
python3 - <<PY
import csv, collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open('/tmp/prof_serial/r_kernel_trace.csv')):
    agg[(r['Kernel_Name'][:80], r['Grid_Size_X'], r['Workgroup_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
with open('$R/$O/serial_by_grid.csv','w') as f:
    w=csv.writer(f); w.writerow(['kernel','grid_x','wg_x','dispatches','mean_us','min_us','total_us'])
    for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
        w.writerow([k[0],k[1],k[2],len(v),round(sum(v)/len(v),1),round(min(v),1),round(sum(v),1)])
PY
head -30 $R/$O/serial_by_grid.csv
