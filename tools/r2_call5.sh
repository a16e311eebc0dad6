#!/bin/bash
# Round-2 GPU call 5: fused weight gradient v3 (mirror ring, 4-instruction items; 8-wave in-workgroup split-K form)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c5; mkdir -p $O; export TMPDIR=/tmp
( time timeout 600 python -m pytest tests -m gpu -q -k "weight_gradient or fused or stem or train_step_fp32_parity_small or fp16_close" ) > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log; grep -E "^FAILED|^ERROR" $O/gpu_tests.log | head
timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "^stem" | tee $O/conv_bench_stem.txt
for w in 8 4; do echo "--- MN_WGF_WAVES=$w"; MN_WGF_WAVES=$w CB_MATCH="3x3" timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "^layer" | grep -v "halo kernel" | sed -e 's/fwd.*wgrad/wgrad(atomics)/'; done | tee $O/conv_bench_waves.txt
for pd in 2 6; do echo "--- MN_WGF_PD=$pd"; MN_WGF_PD=$pd CB_MATCH="3x3" timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "workspace"; done | tee $O/conv_bench_pd.txt
for b in 512 384; do echo "--- MN_WGF_BLOCKS=$b"; MN_WGF_BLOCKS=$b CB_MATCH="3x3" timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "workspace"; done | tee $O/conv_bench_blocks.txt
A=$GRAFT_REPO_ROOT/tools/ablation/libmapnet_hip_abl.so
for a in 0 1 2 4 8 9; do echo "--- fused wgrad ablation $a"; MN_LIB=$A MN_WGF_ABLATE=$a CB_MATCH="3x3" timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "workspace"; done | tee $O/wgf_ablation.txt
timeout 900 bash tools/ab.sh "MN_X=0" "MN_WGRAD_FUSED=0" "MN_WGF_WAVES=4" "MN_WGF_WS=0" "MN_STEM_KERNEL=0" > $O/ab.txt 2>&1; cat $O/ab.txt
TAG=r2c5 BENCH_ARGS="--no-cpu-baseline" timeout 900 bash tools/gpu_prof.sh
