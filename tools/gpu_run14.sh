#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for p in 0 1; do for c in 1 2; do echo "== prio $p config $c"; MN_PRIO=$p MN_IGEMM_CONFIG=$c timeout 300 python tools/conv_bench.py fp16 2>&1 | grep "plain GEMM"; done; done
