#!/bin/bash
# Round-2 GPU call 16: 128 x 64 register tiles for layer2 (igemm_rt.h)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c16; mkdir -p $O; export TMPDIR=/tmp
( time timeout 600 python -m pytest tests -m gpu -q -k "igemm_rt or conv_halo" ) > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log; grep -E "^FAILED|^ERROR" $O/gpu_tests.log | head
timeout 300 python tools/conv_bench.py fp16 192 2>&1 | grep -E "^layer[234] 3x3 [0-9]+->[0-9]+ " | tee $O/conv_bench_rt.txt
timeout 900 bash tools/ab.sh "MN_HALO_PP=1" "MN_HALO_PP=1 MN_IGEMM_RT=1" "MN_HALO_PP=1 MN_IGEMM_RT=2" > $O/ab.txt 2>&1; cat $O/ab.txt
( time MN_HALO_PP=1 MN_IGEMM_RT=2 timeout 900 python -m pytest tests -m gpu -q -x -k "train_step or full_size" ) > $O/gpu_tests_rt.log 2>&1; tail -3 $O/gpu_tests_rt.log
