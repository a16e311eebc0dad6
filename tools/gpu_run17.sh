#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
echo "== normal"; timeout 300 python tools/conv_bench.py fp16 2>&1 | grep -v "amdgpu.ids\|^dtype\|GEMM" | sed 's/.*| wgrad/wgrad/'
echo "== no atomic epilogue"; MN_WGRAD_NOSTORE=1 timeout 300 python tools/conv_bench.py fp16 2>&1 | grep -v "amdgpu.ids\|^dtype\|GEMM" | sed 's/.*| wgrad/wgrad/'
