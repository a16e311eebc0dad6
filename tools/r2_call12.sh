#!/bin/bash
# Round-2 GPU call 12: persistent two-group layer1 convolution (halo_pp.h)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c12; mkdir -p $O; export TMPDIR=/tmp
( time timeout 600 python -m pytest tests -m gpu -q -k "conv_halo" ) > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log; grep -E "^FAILED|^ERROR" $O/gpu_tests.log | head
for side in 1 0 2; do
  echo "== MN_HALO_PP_SIDE=$side"
  MN_HALO_PP_SIDE=$side CB_MATCH="layer1" CB_PP_WGS="0,128" timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "halo"
done | tee $O/conv_bench_pp.txt
timeout 900 bash tools/ab.sh "MN_HALO_PP=0" "MN_HALO_PP=1" "MN_HALO_PP=1 MN_HALO_PP_SIDE=0" > $O/ab.txt 2>&1; cat $O/ab.txt
( time MN_HALO_PP=1 timeout 900 python -m pytest tests -m gpu -q -x -k "train_step or full_size" ) > $O/gpu_tests_pp.log 2>&1; tail -3 $O/gpu_tests_pp.log
