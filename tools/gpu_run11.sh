#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
MN_IGEMM_CONFIG=2 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "conv_forward or conv_data or stem or adjoint or train_step_fp32_parity_small" 2>&1 | grep -E "^FAILED|Error|assert|passed|failed" | head -20
