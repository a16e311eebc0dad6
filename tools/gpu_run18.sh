#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for b in 320 512 768 1536; do echo "== wgrad blocks $b"; MN_WGRAD_BLOCKS=$b timeout 300 python tools/conv_bench.py fp16 2>&1 | grep -v "amdgpu.ids\|^dtype\|GEMM" | sed 's/.*| wgrad/wgrad/' | tr '\n' ';'; echo; done
