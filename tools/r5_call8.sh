#!/bin/bash
# round 5, GPU call 8: the B operand straight from L2 (igemm_halo BG) per launch
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c8; mkdir -p $O
for rep in 1 2; do
for arm in "X=0" "MN_HALO_BG=7"; do echo "== fp16 $arm" | tee -a $O/bg.txt; env $arm python tools/conv_bench.py fp16 2>&1 | grep -E "^layer[234] 3x3 [0-9]+->[0-9]+ +M=" | cut -c1-200 | tee -a $O/bg.txt; done
done
