#!/bin/bash
# Round-2 GPU call 15: halo_pp fragment read-ahead (2 / 3 sub-steps) and early residual / gate requests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c15; mkdir -p $O; export TMPDIR=/tmp
for abl in 200 201 210 211 200 211; do
  echo "== variant $abl (2x0: PD=2, 21x: PD=3; xx1: early side loads)"
  MN_LIB=tools/ablation/libmapnet_hip_abl.so MN_HALO_PP_ABLATE=$abl CB_MATCH="layer1" timeout 100 python tools/conv_bench.py fp16 192 2>&1 | grep -E "halo_pp"
done | tee $O/pp_variants.txt
( time timeout 600 python -m pytest tests -m gpu -q -k "conv_halo" ) > $O/gpu_tests.log 2>&1; tail -2 $O/gpu_tests.log
