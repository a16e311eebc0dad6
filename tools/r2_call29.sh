#!/bin/bash
# Round-2 GPU call 29: halo_pp tile shape, same box: 8x32 (conflict-free fragment reads) vs 16x16
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c29; mkdir -p $O; export TMPDIR=/tmp
for tile in 32 16 32 16; do
  echo "== MN_HALO_PP_TILE=$tile"
  MN_HALO_PP_TILE=$tile CB_MATCH="layer1" timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "halo_pp"
done | tee $O/conv_bench_pp_tile.txt
timeout 600 bash tools/ab.sh "MN_HALO_PP_TILE=32" "MN_HALO_PP_TILE=16" > $O/ab.txt 2>&1; cat $O/ab.txt
( MN_HALO_PP_TILE=16 timeout 300 python -m pytest tests -m gpu -q -k "conv_halo_pp" ) 2>&1 | grep -E "passed|failed" | tail -1
