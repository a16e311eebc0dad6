#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for c in 2 6 7; do echo "== config $c"; MN_IGEMM_CONFIG=$c timeout 300 python tools/conv_bench.py fp16 2>&1 | grep -v "amdgpu.ids\|^dtype\|stem" | sed 's/M= *[0-9]* N= *[0-9]* K= *[0-9]* //; s/(io[^)]*)//; s/| wgrad.*//'; done
