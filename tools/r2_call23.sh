#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c23; mkdir -p $O; export TMPDIR=/tmp
( GA_ALL=1 timeout 600 python tools/grad_accuracy.py fp32 2 64 85 ) 2>&1 | grep -v Warning | tee $O/grad_accuracy_fp32.txt
