#!/bin/bash
# Round-2 GPU call 20: the four tests whose first-run tolerances were adjusted after call 19
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c20; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -k "second_step or sgd or rmsprop or fused_adam" ) > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log; grep -E "^FAILED|^ERROR|Error" $O/gpu_tests.log | head
