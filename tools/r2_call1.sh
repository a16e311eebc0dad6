#!/bin/bash
# Round-2 GPU call 1: suite (with the new full-size parity / NaN-filter / overflow tests), timing of everything written
# but untimed at the end of round 1, fp16 + fp32 bench records, per-layer error table, rocprofv3 stats.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c1; mkdir -p $O; export TMPDIR=/tmp
rm -f gpurun_out/parity_full_size.jsonl
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
cp gpurun_out/parity_full_size.jsonl $O/ 2>/dev/null
timeout 200 python tools/conv_bench.py fp16 192 > $O/conv_bench_default.txt 2>&1
MN_WGRAD_TR_ASM=1 CB_MATCH="3x3" timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "layer" > $O/conv_bench_tr_asm.txt
MN_IGEMM_HALO=2 CB_MATCH="3x3" timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "layer" > $O/conv_bench_halo2.txt
echo "--- default"; cat $O/conv_bench_default.txt; echo "--- MN_WGRAD_TR_ASM=1"; cat $O/conv_bench_tr_asm.txt; echo "--- MN_IGEMM_HALO=2"; cat $O/conv_bench_halo2.txt
timeout 900 bash tools/ab.sh "MN_X=0" "MN_WGRAD_TR_ASM=1" "MN_IGEMM_HALO=1" "MN_IGEMM_HALO=2" "MN_FORCE_STAGED=1" "MN_BN_BWD_REVERSE=1" "MN_BN_REDUCE_BLOCKS=1024" "MN_OVERFLOW_GUARD=0" > $O/ab.txt 2>&1; cat $O/ab.txt
timeout 600 python bench.py --steps 50 --warmup 5 > $O/bench_fp16.json 2> $O/bench_fp16.err; tail -1 $O/bench_fp16.json
timeout 600 python bench.py --steps 20 --warmup 3 --dtype fp32 > $O/bench_fp32.json 2> $O/bench_fp32.err; tail -1 $O/bench_fp32.json
timeout 600 python tools/layer_error.py > $O/layer_error.txt 2>&1; tail -70 $O/layer_error.txt
cp gpurun_out/layer_error.json $O/ 2>/dev/null
TAG=r2c1 BENCH_ARGS="--no-cpu-baseline" timeout 900 bash tools/gpu_prof.sh
