#!/bin/bash
# Round-2 GPU call 4: fused weight gradient v2 (fragment prefetch PD, workspace partials + reduce launch)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c4; mkdir -p $O; export TMPDIR=/tmp
( time timeout 600 python -m pytest tests -m gpu -q -k "nan_filter or weight_gradient or fused or train_step_fp32_parity_small or fp16_close" ) > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log; grep -E "^FAILED|^ERROR" $O/gpu_tests.log | head
for pd in 4 1 2 6; do echo "--- MN_WGF_PD=$pd"; MN_WGF_PD=$pd CB_MATCH="3x3" timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "^layer" | grep -v "halo kernel" | sed -e 's/fwd.*wgrad/wgrad(atomics)/'; done | tee $O/conv_bench_pd.txt
for b in 256 384; do echo "--- MN_WGF_BLOCKS=$b"; MN_WGF_BLOCKS=$b CB_MATCH="3x3" timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "workspace"; done | tee $O/conv_bench_blocks.txt
A=$GRAFT_REPO_ROOT/tools/ablation/libmapnet_hip_abl.so
for a in 0 1 4 8; do echo "--- fused wgrad ablation $a"; MN_LIB=$A MN_WGF_ABLATE=$a CB_MATCH="3x3" timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "workspace|^layer[134] " | grep -v "halo kernel"| sed -e 's/fwd.*wgrad/wgrad(atomics)/'; done | tee $O/wgf_ablation.txt
timeout 900 bash tools/ab.sh "MN_X=0" "MN_WGRAD_FUSED=0" "MN_WGF_WS=0" "MN_WGF_BLOCKS=256" "MN_BN_REDUCE_BLOCKS=512" > $O/ab.txt 2>&1; cat $O/ab.txt
TAG=r2c4 BENCH_ARGS="--no-cpu-baseline" timeout 900 bash tools/gpu_prof.sh
