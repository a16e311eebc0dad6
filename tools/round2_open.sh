#!/bin/bash
# First GPU call of a round: everything the end-of-round-1 changes still need measured, in one gpurun.
#   gpurun --timeout 1500 -- 'bash tools/round2_open.sh'
# 1. GPU suite (includes the forced 288x256 cases and the MN_WGRAD_TR_ASM race screen).
# 2. Per-layer kernel rates (layer3 rows show the effect of the scratch-reload fix; compare with profiles/r01).
# 3. Weight-gradient rates and whole-step A/B with the assembly transpose reads on / off.
# 4. BatchNorm-backward reduction: workgroup count A/B (MN_BN_REDUCE_BLOCKS).
# 5. Bench + rocprofv3 stats of the default tree; LDS-DMA rate probe.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2_open; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 300 python tools/conv_bench.py fp16 192 > $O/conv_bench_default.txt 2>&1
MN_WGRAD_TR_ASM=1 timeout 300 python tools/conv_bench.py fp16 192 > $O/conv_bench_tr_asm.txt 2>&1
echo "--- default"; cat $O/conv_bench_default.txt; echo "--- MN_WGRAD_TR_ASM=1"; grep wgrad $O/conv_bench_tr_asm.txt
timeout 900 bash tools/ab.sh "MN_WGRAD_TR_ASM=0" "MN_WGRAD_TR_ASM=1" "MN_WGRAD_TR_ASM=1 MN_WGRAD_VARIANT=0" > $O/ab_tr_asm.txt 2>&1; cat $O/ab_tr_asm.txt
MN_IGEMM_HALO=2 timeout 300 python tools/conv_bench.py fp16 192 > $O/conv_bench_halo.txt 2>&1; echo "--- MN_IGEMM_HALO=2 (layers 2-4)"; cat $O/conv_bench_halo.txt
timeout 900 bash tools/ab.sh "MN_IGEMM_HALO=0" "MN_IGEMM_HALO=1" "MN_IGEMM_HALO=2" > $O/ab_halo.txt 2>&1; cat $O/ab_halo.txt
timeout 900 bash tools/ab.sh "MN_BN_REDUCE_BLOCKS=4096" "MN_BN_REDUCE_BLOCKS=1024" "MN_BN_REDUCE_BLOCKS=512" > $O/ab_bn_blocks.txt 2>&1; cat $O/ab_bn_blocks.txt
timeout 400 bash tools/ab.sh "MN_FORCE_STAGED=0" "MN_FORCE_STAGED=1" > $O/ab_staged.txt 2>&1; cat $O/ab_staged.txt   # cost of the data-parallel (staged) issue order on one GPU
TAG=r2_open timeout 1200 bash tools/gpu_prof.sh
timeout 120 tools/probes/dma_probe > $O/dma_probe.txt 2>&1; cat $O/dma_probe.txt
