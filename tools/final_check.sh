#!/bin/bash
# End-of-round check on the GPU box (gpurun -- 'TAG=cNN bash tools/final_check.sh'): the default bench line, serial kernel stats
# of the fp16 step and of both parity modes (fp16x2m, fp16x2), PMC passes (FETCH_SIZE / WRITE_SIZE) of the fp16 and fp16x2m steps,
# the full GPU suite and smoke();
# everything lands in gpurun_out/final_$TAG.  SKIP_SUITE=1 / SKIP_PMC=1 shorten it.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_${TAG:-cur}; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench.err; tail -1 $O/bench_default.json | cut -c1-300
cd /tmp && MN_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_serial -o r -- python $R/bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-events --no-parity-mode --no-eval-metric > $O/rocprof.log 2>&1
cp /tmp/prof_serial/r_kernel_stats.csv $O/kernel_stats_serial.csv
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ovl -o r -- python $R/bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-events --no-parity-mode --no-eval-metric >> $O/rocprof.log 2>&1
cp /tmp/prof_ovl/r_kernel_stats.csv $O/kernel_stats_overlapped.csv
for dt in fp16x2m fp16x2; do
cd /tmp && MN_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$dt -o r -- python $R/bench.py --dtype $dt --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --no-events --no-parity-mode --no-eval-metric >> $O/rocprof.log 2>&1
cp /tmp/prof_$dt/r_kernel_stats.csv $O/kernel_stats_serial_$dt.csv
done
if [ -z "$SKIP_PMC" ]; then
mkdir -p $R/gpurun_out/prof_final_${TAG:-cur}
for dt in fp16 fp16x2m; do
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && rm -rf /tmp/prof_$c && MN_WGRAD_STREAM=0 timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prof_$c -o r -- python $R/bench.py --dtype $dt --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-events --no-parity-mode --no-eval-metric > /dev/null 2>&1
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
  cd $R && python3 tools/pmc_conv_traffic.py $O $O "round 5, fp16 step (tools/final_check.sh, TAG=${TAG:-cur})" > /dev/null 2>&1
else
  mkdir -p /tmp/pmc_$dt && cp $O/pmc_FETCH_SIZE_$dt.csv /tmp/pmc_$dt/pmc_FETCH_SIZE.csv && cp $O/pmc_WRITE_SIZE_$dt.csv /tmp/pmc_$dt/pmc_WRITE_SIZE.csv
  cd $R && python3 tools/pmc_conv_traffic.py /tmp/pmc_$dt $O "round 5, $dt step (tools/final_check.sh, TAG=${TAG:-cur})" _$dt > /dev/null 2>&1
fi
done
fi
if [ -z "$SKIP_SUITE" ]; then
cd $R && timeout 2700 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tee $O/gpu_suite_summary.txt
fi
cd $R && timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8 | tee $O/smoke.txt
