#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for c in 1 2 3 4; do echo "== igemm config $c"; MN_IGEMM_CONFIG=$c timeout 300 python tools/conv_bench.py fp16 2>&1 | grep "plain GEMM"; done
