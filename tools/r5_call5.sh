#!/bin/bash
# round 5, GPU call 5: new tests; how much the second stream hides in fp16x2m; BatchNorm-reduce rows in flight for the mixed-type kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "trajectory or staged or trains_to_the_accuracy" 2>&1 | tail -5 | tee $O/tests.txt
cp gpurun_out/loss_trajectory_fp16x2m_vs_fp16x2.json gpurun_out/accuracy_fp16_vs_fp16x2m_five_seeds.json $O/ 2>/dev/null
run() { env $2 python bench.py --dtype $1 --no-cpu-baseline --no-events --no-parity-mode --no-eval-metric --steps 30 --warmup 8 --repeats 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  for arm in "X=0" "MN_WGRAD_STREAM=0" "MN_BN_REDUCE_U=4"; do
    echo "[fp16x2m $arm] $(run fp16x2m "$arm")" | tee -a $O/ab.txt
  done
done
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ovl -o r -- python $R/bench.py --dtype fp16x2m --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --no-events --no-parity-mode --no-eval-metric > $R/$O/rocprof.log 2>&1
cp /tmp/prof_ovl/r_kernel_stats.csv $R/$O/kernel_stats_overlapped_fp16x2m.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/prof_ovl/r_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
st=[i for i,n in enumerate(names) if 'nchw_to_padded_nhwc4_kernelIf' in n or 'nchw_to_padded_nhwc4_kernel<float>' in n]
s,e=st[-2],st[-1]
t0=int(rows[s]['Start_Timestamp'])
with open('$R/$O/one_step_trace_fp16x2m.csv','w') as f:
    w=csv.writer(f); w.writerow(['start_us','dur_us','queue','grid','kernel'])
    for r in rows[s:e]:
        w.writerow([round((int(r['Start_Timestamp'])-t0)/1e3,1), round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,1), r.get('Queue_Id',''), r['Grid_Size_X'], r['Kernel_Name'][:100]])
print('step wall us', (int(rows[e]['Start_Timestamp'])-t0)/1e3, 'sum of kernel durations us', sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows[s:e])/1e3)
PY
