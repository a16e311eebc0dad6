#!/bin/bash
# Round-2 GPU call 8: stem backward in two launches
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c11; mkdir -p $O; export TMPDIR=/tmp
( time timeout 600 python -m pytest tests -m gpu -q -k "stem or train_step_fp32_parity_small or fp16_close or finite_and_learns" ) > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log; grep -E "^FAILED|^ERROR" $O/gpu_tests.log | head
timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "^stem" | tee $O/conv_bench_stem.txt
timeout 900 bash tools/ab.sh "MN_X=0" "MN_STEM_BWD=0" > $O/ab.txt 2>&1; cat $O/ab.txt
TAG=r2c11 BENCH_ARGS="--no-cpu-baseline" timeout 900 bash tools/gpu_prof.sh
