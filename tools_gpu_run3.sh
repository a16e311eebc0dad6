#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== gpu tests"
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/gpu_tests.log 2>&1
tail -5 gpurun_out/gpu_tests.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== bench"
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1
tail -1 gpurun_out/bench.log
echo "== rocprof stats"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o r1 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/rocprof_stats.log 2>&1
echo "rc=$?"; tail -2 $R/gpurun_out/rocprof_stats.log | cut -c1-400
mkdir -p $R/gpurun_out/prof_r1; find /tmp/prof_stats -name "*stats*.csv" -exec cp {} $R/gpurun_out/prof_r1/ \;
find /tmp/prof_stats -name "*kernel_trace.csv" -exec sh -c 'head -2000 "$1" > '$R'/gpurun_out/prof_r1/kernel_trace_head.csv' _ {} \;
ls -la /tmp/prof_stats/* | head
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c"
  cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prof_$c -o r1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-events > $R/gpurun_out/rocprof_$c.log 2>&1
  echo "rc=$?"
  find /tmp/prof_$c -name "*counter_collection.csv" -exec cp {} $R/gpurun_out/prof_r1/pmc_$c.csv \;
done
ls -la $R/gpurun_out/prof_r1
