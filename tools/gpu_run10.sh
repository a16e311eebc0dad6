#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for c in 1 2 3 4 5; do echo "== igemm config $c"; MN_IGEMM_CONFIG=$c timeout 300 python tools/conv_bench.py fp16 2>&1 | grep -v "amdgpu.ids\|^dtype" | sed 's/M= *[0-9]* N= *[0-9]* K= *[0-9]* //; s/(io[^)]*)//; s/| wgrad.*//' ; done
echo "== correctness"; for c in 2 3 4 5; do MN_IGEMM_CONFIG=$c timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "conv_forward or conv_data or stem or adjoint or train_step_fp32_parity_small" 2>&1 | tail -1; done
