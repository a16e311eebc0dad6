#!/bin/bash
# Round-2 GPU call 25: memory-level parallelism of the BatchNorm backward passes (whole-step A/B)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c25; mkdir -p $O; export TMPDIR=/tmp
timeout 900 bash tools/ab.sh "MN_X=0" "MN_BN_REDUCE_UNROLL=8" "MN_BN_REDUCE_UNROLL=8 MN_BN_REDUCE_BLOCKS=1024" "MN_EW_WGS_PER_CU=32" "MN_EW_WGS_PER_CU=8" > $O/ab.txt 2>&1; cat $O/ab.txt
