#!/bin/bash
# Round-2 GPU call 14: halo_pp ablations (which phase bounds the 5 us per tile pair?) and wave priority
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c14; mkdir -p $O; export TMPDIR=/tmp
for abl in 0 1 2 4 5 8 10 13 100 101; do
  echo "== MN_HALO_PP_ABLATE=$abl"
  MN_LIB=tools/ablation/libmapnet_hip_abl.so MN_HALO_PP_ABLATE=$abl CB_MATCH="layer1" timeout 100 python tools/conv_bench.py fp16 192 2>&1 | grep -E "halo_pp"
done | tee $O/pp_ablation.txt
