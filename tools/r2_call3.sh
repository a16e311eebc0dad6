#!/bin/bash
# Round-2 GPU call 3: fused weight gradient with 1-3 steps in flight, ablations of the fused wgrad and of the layer1 halo
# kernel (ablation build), BatchNorm block counts.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c3; mkdir -p $O; export TMPDIR=/tmp
( time timeout 600 python -m pytest tests -m gpu -q -k "nan_filter or weight_gradient or fused or conv_halo" ) > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log; grep -E "^FAILED|^ERROR" $O/gpu_tests.log | head
for d in 1 2 3; do echo "--- MN_WGF_DEPTH=$d"; MN_WGF_DEPTH=$d CB_MATCH="3x3" timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "^layer" | sed -e 's/fwd.*wgrad/wgrad/'; done | tee $O/conv_bench_depth.txt
A=$GRAFT_REPO_ROOT/tools/ablation/libmapnet_hip_abl.so
for a in 0 1 2 4 8 3 6 11; do echo "--- fused wgrad ablation $a"; MN_LIB=$A MN_WGF_ABLATE=$a CB_MATCH="3x3" timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "^layer[134] " | sed -e 's/fwd.*wgrad/wgrad/'; done | tee $O/wgf_ablation.txt
for a in 0 1 2 4 8 3 7 12 15; do echo "--- halo ablation $a"; MN_LIB=$A MN_HALO_ABLATE=$a CB_MATCH="layer1" timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "halo kernel"; done | tee $O/halo_ablation.txt
timeout 900 bash tools/ab.sh "MN_X=0" "MN_WGRAD_FUSED=0" "MN_WGF_DEPTH=2" "MN_BN_REDUCE_BLOCKS=512" "MN_BN_REDUCE_BLOCKS=256" "MN_WGRAD_SCHED=0" > $O/ab.txt 2>&1; cat $O/ab.txt
