#!/bin/bash
# Round-2 GPU call 13: halo_pp v2 (register epilogue through swapped MFMA operands, halo DMA issued before the stores)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c13; mkdir -p $O; export TMPDIR=/tmp
( time timeout 600 python -m pytest tests -m gpu -q -k "conv_halo" ) > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log; grep -E "^FAILED|^ERROR" $O/gpu_tests.log | head
CB_MATCH="layer1" CB_PP_WGS="0,255,192" timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "halo" | tee $O/conv_bench_pp.txt
timeout 900 bash tools/ab.sh "MN_HALO_PP=0" "MN_HALO_PP=1" > $O/ab.txt 2>&1; cat $O/ab.txt
( time MN_HALO_PP=1 timeout 900 python -m pytest tests -m gpu -q -x -k "train_step or full_size" ) > $O/gpu_tests_pp.log 2>&1; tail -3 $O/gpu_tests_pp.log
