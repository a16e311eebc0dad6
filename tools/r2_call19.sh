#!/bin/bash
# Round-2 GPU call 19 (final): full GPU suite, bench lines (fp16 with parity + CPU baseline, fp32), rocprofv3 stats + PMC,
# weight-gradient schedule A/B on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c19; mkdir -p $O; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/gpu_suite.log 2>&1; tail -5 $O/gpu_suite.log; grep -E "^FAILED|^ERROR" $O/gpu_suite.log | head -20
timeout 600 python bench.py > $O/bench_fp16.json 2> $O/bench_fp16.err; tail -1 $O/bench_fp16.json
timeout 600 python bench.py --dtype fp32 --steps 10 --warmup 3 > $O/bench_fp32.json 2> $O/bench_fp32.err; tail -1 $O/bench_fp32.json
TAG=r2c19 PMC=1 BENCH_ARGS="--no-cpu-baseline" timeout 1200 bash tools/gpu_prof.sh
timeout 600 bash tools/ab.sh "MN_WGRAD_SCHED=2" "MN_WGRAD_SCHED=0" "MN_WGRAD_SCHED=1" > $O/ab_sched.txt 2>&1; cat $O/ab_sched.txt
