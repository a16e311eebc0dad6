#!/bin/bash
# bench + rocprofv3 kernel stats (+ optional PMC passes with PMC=1); summaries land in gpurun_out/prof_$TAG
#   kernel_stats.csv         default execution (weight-gradient launches overlap the data-gradient chain)
#   serial_by_grid.csv       the serial run's dispatches averaged per (kernel, grid): per-layer durations
#   kernel_stats_serial_fp32x3.csv  (X3=1) the parity mode, serial
#   kernel_stats_serial.csv  MN_WGRAD_STREAM=0: every kernel alone on the device (durations comparable with the
#                            HIP-event timings bench.py reports in `roofline`)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${TAG:-cur}; R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_$TAG; export TMPDIR=/tmp
timeout 900 python bench.py ${BENCH_ARGS} > gpurun_out/prof_$TAG/bench.json 2> gpurun_out/prof_$TAG/bench.err; tail -1 gpurun_out/prof_$TAG/bench.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o r -- python $R/bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-events --no-parity-mode --no-eval-metric > $R/gpurun_out/prof_$TAG/rocprof.log 2>&1
cp /tmp/prof_stats/r_kernel_stats.csv $R/gpurun_out/prof_$TAG/kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/prof_stats/r_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
st=[i for i,n in enumerate(names) if 'nchw_to_padded' in n]
s,e=st[-2],st[-1]
with open('$R/gpurun_out/prof_$TAG/one_step_trace.csv','w') as f:
    w=csv.writer(f); w.writerow(['start_us','dur_us','grid','kernel'])
    t0=int(rows[s]['Start_Timestamp'])
    for r in rows[s:e]:
        w.writerow([round((int(r['Start_Timestamp'])-t0)/1e3,1), round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,1), r['Grid_Size_X'], r['Kernel_Name'][:110]])
print('step wall us', (int(rows[e]['Start_Timestamp'])-int(rows[s]['Start_Timestamp']))/1e3)
PY
cd /tmp && MN_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_serial -o r -- python $R/bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-events --no-parity-mode --no-eval-metric >> $R/gpurun_out/prof_$TAG/rocprof.log 2>&1
cp /tmp/prof_serial/r_kernel_stats.csv $R/gpurun_out/prof_$TAG/kernel_stats_serial.csv
python3 - <<PY
import csv, collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open('/tmp/prof_serial/r_kernel_trace.csv')):
    agg[(r['Kernel_Name'][:80], r['Grid_Size_X'], r['Workgroup_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
with open('$R/gpurun_out/prof_$TAG/serial_by_grid.csv','w') as f:
    w=csv.writer(f); w.writerow(['kernel','grid_x','wg_x','dispatches','mean_us','min_us','total_us'])
    for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
        w.writerow([k[0],k[1],k[2],len(v),round(sum(v)/len(v),1),round(min(v),1),round(sum(v),1)])
PY
if [ -n "$X3" ]; then
cd /tmp && MN_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x3 -o r -- python $R/bench.py --dtype fp32x3 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-events --no-eval-metric >> $R/gpurun_out/prof_$TAG/rocprof.log 2>&1
cp /tmp/prof_x3/r_kernel_stats.csv $R/gpurun_out/prof_$TAG/kernel_stats_serial_fp32x3.csv
fi
if [ -n "$PMC" ]; then for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && MN_WGRAD_STREAM=0 timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prof_$c -o r -- python $R/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-events --no-parity-mode --no-eval-metric > /dev/null 2>&1
  python3 - <<PY
import csv, collections
agg=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open('/tmp/prof_$c/r_counter_collection.csv')):
    k=r['Kernel_Name'][:90]; agg[k][0]+=1; agg[k][1]+=float(r['Counter_Value'])
with open('$R/gpurun_out/prof_$TAG/pmc_$c.csv','w') as f:
    w=csv.writer(f); w.writerow(['kernel','dispatches','sum_$c','per_dispatch'])
    for k,(n,v) in sorted(agg.items(), key=lambda kv:-kv[1][1]): w.writerow([k,n,v,v/n])
PY
done; fi
