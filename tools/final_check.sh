#!/bin/bash
# End-of-round check on the GPU box (gpurun -- 'TAG=cNN bash tools/final_check.sh'): the default bench line (fp16x2m = the number of
# record, fp16 beside it), serial kernel stats of both steps, PMC passes (FETCH_SIZE / WRITE_SIZE) and SQ counter passes
# (tools/sq_counters.sh) of both, the full GPU suite and smoke(); everything lands in gpurun_out/final_$TAG.
# SKIP_SUITE=1 / SKIP_PMC=1 / SKIP_SQ=1 shorten it.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_${TAG:-cur}; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench.err; tail -1 $O/bench_default.json | cut -c1-300
LEAN="--steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-events --no-fast-mode --no-eval-metric --no-feed"
for dt in fp16x2m fp16; do
cd /tmp && MN_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$dt -o r -- python $R/bench.py --dtype $dt $LEAN >> $O/rocprof.log 2>&1
cp /tmp/prof_$dt/r_kernel_stats.csv $O/kernel_stats_serial_$dt.csv
python3 - <<PY
import csv, collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open('/tmp/prof_$dt/r_kernel_trace.csv')):
    agg[(r['Kernel_Name'][:100], r['Grid_Size_X'], r['Workgroup_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
with open('$O/serial_by_grid_$dt.csv','w') as f:
    w=csv.writer(f); w.writerow(['kernel','grid_x','wg_x','dispatches','mean_us','min_us','total_us'])
    for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
        w.writerow([k[0],k[1],k[2],len(v),round(sum(v)/len(v),1),round(min(v),1),round(sum(v),1)])
PY
done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ovl -o r -- python $R/bench.py $LEAN >> $O/rocprof.log 2>&1
cp /tmp/prof_ovl/r_kernel_stats.csv $O/kernel_stats_overlapped_fp16x2m.csv
if [ -z "$SKIP_PMC" ]; then
for dt in fp16x2m fp16; do
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && rm -rf /tmp/prof_$c && MN_WGRAD_STREAM=0 timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prof_$c -o r -- python $R/bench.py --dtype $dt --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-events --no-fast-mode --no-eval-metric --no-feed > /dev/null 2>&1
  python3 - <<PY
import csv, collections
agg=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open('/tmp/prof_$c/r_counter_collection.csv')):
    k=r['Kernel_Name'][:90]; agg[k][0]+=1; agg[k][1]+=float(r['Counter_Value'])
with open('$O/pmc_$c' + ('' if '$dt' == 'fp16' else '_$dt') + '.csv','w') as f:
    w=csv.writer(f); w.writerow(['kernel','dispatches','sum_$c','per_dispatch'])
    for k,(n,v) in sorted(agg.items(), key=lambda kv:-kv[1][1]): w.writerow([k,n,v,v/n])
PY
done
if [ $dt = fp16 ]; then
  cd $R && python3 tools/pmc_conv_traffic.py $O $O "round 6, fp16 step (tools/final_check.sh, TAG=${TAG:-cur})" > /dev/null 2>&1
else
  mkdir -p /tmp/pmc_$dt && cp $O/pmc_FETCH_SIZE_$dt.csv /tmp/pmc_$dt/pmc_FETCH_SIZE.csv && cp $O/pmc_WRITE_SIZE_$dt.csv /tmp/pmc_$dt/pmc_WRITE_SIZE.csv
  cd $R && python3 tools/pmc_conv_traffic.py /tmp/pmc_$dt $O "round 6, $dt step (tools/final_check.sh, TAG=${TAG:-cur})" _$dt > /dev/null 2>&1
fi
done
fi
if [ -z "$SKIP_SQ" ]; then
cd $R && TAG=final_${TAG:-cur} bash tools/sq_counters.sh > $O/sq.log 2>&1; cp $R/gpurun_out/sq_final_${TAG:-cur}/sq_counters_*.{txt,json} $O/ 2>/dev/null
fi
if [ -z "$SKIP_SUITE" ]; then
cd $R && timeout 2700 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tee $O/gpu_suite_summary.txt
fi
cd $R && timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8 | tee $O/smoke.txt
