#!/bin/bash
# Round-2 GPU call 2: new defaults (chunk-resident A kernel, fused weight gradient, BatchNorm finalize launches) --
# full GPU suite, per-layer rates, A/Bs against the previous forms, profile.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c2; mkdir -p $O; export TMPDIR=/tmp
rm -f gpurun_out/parity_full_size.jsonl
( time timeout 1200 python -m pytest tests -m gpu -q ) > $O/gpu_tests.log 2>&1; tail -6 $O/gpu_tests.log; grep -E "^FAILED|^ERROR" $O/gpu_tests.log | head
cp gpurun_out/parity_full_size.jsonl $O/ 2>/dev/null
CB_MATCH="3x3" timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "layer" > $O/conv_bench_default.txt
MN_WGRAD_FUSED=0 CB_MATCH="3x3" timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "layer" > $O/conv_bench_unfused.txt
echo "--- default (fused wgrad)"; cat $O/conv_bench_default.txt; echo "--- MN_WGRAD_FUSED=0"; cat $O/conv_bench_unfused.txt
timeout 900 bash tools/ab.sh "MN_X=0" "MN_WGRAD_FUSED=0" "MN_IGEMM_HALO=0" "MN_BN_REDUCE_BLOCKS=4096" "MN_BN_REDUCE_BLOCKS=512" "MN_EW_WGS_PER_CU=8" "MN_WGRAD_SCHED=0" > $O/ab.txt 2>&1; cat $O/ab.txt
TAG=r2c2 BENCH_ARGS="--no-cpu-baseline" timeout 900 bash tools/gpu_prof.sh
