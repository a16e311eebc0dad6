#!/bin/bash
# round 5, GPU call 4: BIG single-image tiles (576 rows, 12 waves of 96 x 64 / 96 x 32) per launch, fp16 and h2
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c4; mkdir -p $O
for rep in 1 2; do
for arm in "X=0" "MN_HALO_BIG=3"; do echo "== fp16 $arm" | tee -a $O/big_tiles.txt; env $arm python tools/conv_bench.py fp16 2>&1 | grep -E "^layer[1234] 3x3 " | cut -c1-200 | tee -a $O/big_tiles.txt; done
for arm in "X=0" "MN_H2_HALO_BIG=7" "MN_H2_HALO_BIG=11"; do echo "== fp16x2 $arm" | tee -a $O/big_tiles.txt; env $arm python tools/conv_bench.py fp16x2 2>&1 | grep -E "^layer[1234] 3x3 " | cut -c1-200 | tee -a $O/big_tiles.txt; done
done
