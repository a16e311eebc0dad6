#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for a in 0 1 2 4 3 5 6 7; do echo -n "ablate=$a (1 noDMA 2 noLDSread 4 noMFMA): "; MN_ABL=$a MN_IGEMM_CONFIG=1 timeout 300 python tools/conv_bench.py fp16 2>&1 | grep "plain GEMM" | head -1; done
