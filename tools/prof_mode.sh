#!/bin/bash
# serial kernel stats of one dtype mode (every kernel alone on the device): DT=fp16x2 TAG=c5 bash tools/prof_mode.sh
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; DT=${DT:-fp16x2}; O=$R/gpurun_out/${TAG:-cur}; mkdir -p $O; export TMPDIR=/tmp
cd /tmp && MN_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o r -- python $R/bench.py --dtype $DT --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --no-events --no-fast-mode --no-eval-metric --no-feed > $O/rocprof_$DT.log 2>&1
cp /tmp/prof_m/r_kernel_stats.csv $O/kernel_stats_serial_$DT.csv
python3 - <<PY
import csv, collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open('/tmp/prof_m/r_kernel_trace.csv')):
    agg[(r['Kernel_Name'][:90], r['Grid_Size_X'], r['Workgroup_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
with open('$O/serial_by_grid_$DT.csv','w') as f:
    w=csv.writer(f); w.writerow(['kernel','grid_x','wg_x','dispatches','mean_us','min_us','total_us'])
    for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
        w.writerow([k[0],k[1],k[2],len(v),round(sum(v)/len(v),1),round(min(v),1),round(sum(v),1)])
PY
