#!/bin/bash
# round 3, GPU call 22: default bench record of the final tree; parity-mode kernel stats (serial); full GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c22; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench.err; tail -1 $O/bench_default.json | cut -c1-400
cd /tmp && MN_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x3 -o r -- python $R/bench.py --dtype fp32x3 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-events > $O/rocprof.log 2>&1
cp /tmp/prof_x3/r_kernel_stats.csv $O/kernel_stats_serial_fp32x3.csv
cd $R && (time timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $O/gpu_suite_tail.txt 2>&1; cat $O/gpu_suite_tail.txt
