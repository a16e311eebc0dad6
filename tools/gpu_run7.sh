#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for b in 256 384 512 768; do echo "== wgrad blocks $b"; MN_WGRAD_BLOCKS=$b timeout 300 python tools/conv_bench.py fp16 2>&1 | grep -v amdgpu.ids | sed 's/.*| wgrad/wgrad/' ; done
echo "== old wgrad kernel, 256 blocks"; MN_WGRAD_DMA=0 MN_WGRAD_BLOCKS=256 timeout 300 python tools/conv_bench.py fp16 2>&1 | grep -v amdgpu.ids | sed 's/.*| wgrad/wgrad/'
