#!/bin/bash
# Round-2 GPU call 6: full suite on the new defaults, persistent stem kernel, scheduling A/Bs, profile + PMC traffic
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c6; mkdir -p $O; export TMPDIR=/tmp
rm -f gpurun_out/parity_full_size.jsonl
( time timeout 1200 python -m pytest tests -m gpu -q ) > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log; grep -E "^FAILED|^ERROR" $O/gpu_tests.log | head
cp gpurun_out/parity_full_size.jsonl $O/ 2>/dev/null
timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "^stem" | tee $O/conv_bench_stem.txt
for w in 256 1024; do echo "--- MN_STEM_WGS=$w"; MN_STEM_WGS=$w timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "^stem kernel"; done | tee -a $O/conv_bench_stem.txt
timeout 900 bash tools/ab.sh "MN_X=0" "MN_STEM_KERNEL=0" "MN_WGRAD_SCHED=0" "MN_WGRAD_SCHED=2" "MN_WGRAD_STREAM=0" "MN_BN_REDUCE_BLOCKS=1024" > $O/ab.txt 2>&1; cat $O/ab.txt
TAG=r2c6 PMC=1 BENCH_ARGS="--no-cpu-baseline" timeout 1200 bash tools/gpu_prof.sh
