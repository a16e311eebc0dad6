#!/bin/bash
# Round-2 GPU call 17: full GPU suite on the tree with halo_pp on, the reordered finalize kernels and MN_DETERMINISTIC;
# whole-step cost of the deterministic mode and of more accumulator rows
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c17; mkdir -p $O; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/gpu_suite.log 2>&1; tail -5 $O/gpu_suite.log; grep -E "^FAILED|^ERROR" $O/gpu_suite.log | head -20
timeout 900 bash tools/ab.sh "MN_X=0" "MN_ACC_ROWS=32" "MN_DETERMINISTIC=1" > $O/ab.txt 2>&1; cat $O/ab.txt
