#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in 1 2 4; do for w in 0 1; do echo "== igemm variant $v wgrad variant $w"; MN_IGEMM_VARIANT=$v MN_WGRAD_VARIANT=$w timeout 300 python tools/conv_bench.py fp16 2>&1 | grep -v "amdgpu.ids\|^dtype" | sed 's/M= *[0-9]* N= *[0-9]* K= *[0-9]* //; s/(io[^)]*)//' ; done; done
