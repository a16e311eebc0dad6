#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 0 1 2 3 4; do
  echo "== conv bench variant $v"
  MN_IGEMM_VARIANT=$v timeout 300 python tools/conv_bench.py fp16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/conv_bench_v$v.log
done
for v in 1 2 3 4; do
  echo "== correctness variant $v"
  MN_IGEMM_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "conv_forward or conv_data or stem or adjoint or train_step_fp32_parity_small" 2>&1 | tail -3
done
