#!/bin/bash
# End-of-round check on the GPU box (gpurun -- 'TAG=cNN bash tools/final_check.sh'): the default bench line, serial kernel stats
# of the fp16 step and of the parity mode, PMC passes (FETCH_SIZE / WRITE_SIZE) of the fp16 step, the full GPU suite and smoke();
# everything lands in gpurun_out/final_$TAG.  SKIP_SUITE=1 / SKIP_PMC=1 shorten it.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_${TAG:-cur}; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench.err; tail -1 $O/bench_default.json | cut -c1-300
cd /tmp && MN_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_serial -o r -- python $R/bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-events --no-parity-mode --no-eval-metric > $O/rocprof.log 2>&1
cp /tmp/prof_serial/r_kernel_stats.csv $O/kernel_stats_serial.csv
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ovl -o r -- python $R/bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-events --no-parity-mode --no-eval-metric >> $O/rocprof.log 2>&1
cp /tmp/prof_ovl/r_kernel_stats.csv $O/kernel_stats_overlapped.csv
cd /tmp && MN_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pm -o r -- python $R/bench.py --dtype fp16x2 --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --no-events --no-parity-mode --no-eval-metric >> $O/rocprof.log 2>&1
cp /tmp/prof_pm/r_kernel_stats.csv $O/kernel_stats_serial_fp16x2.csv
if [ -z "$SKIP_PMC" ]; then
mkdir -p $R/gpurun_out/prof_final_${TAG:-cur}
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && MN_WGRAD_STREAM=0 timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prof_$c -o r -- python $R/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-events --no-parity-mode --no-eval-metric > /dev/null 2>&1
  python3 - <<PY
import csv, collections
agg=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open('/tmp/prof_$c/r_counter_collection.csv')):
    k=r['Kernel_Name'][:90]; agg[k][0]+=1; agg[k][1]+=float(r['Counter_Value'])
with open('$O/pmc_$c.csv','w') as f:
    w=csv.writer(f); w.writerow(['kernel','dispatches','sum_$c','per_dispatch'])
    for k,(n,v) in sorted(agg.items(), key=lambda kv:-kv[1][1]): w.writerow([k,n,v,v/n])
PY
done
cd $R && python3 tools/pmc_conv_traffic.py $O $O "round 4, final tree (tools/final_check.sh, TAG=${TAG:-cur})" > /dev/null 2>&1
fi
if [ -z "$SKIP_SUITE" ]; then
cd $R && timeout 2700 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tee $O/gpu_suite_summary.txt
fi
cd $R && timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8 | tee $O/smoke.txt
