#!/bin/bash
# Round-2 GPU call 21: is the second step's distance from the oracle numerics or scheduling?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c21; mkdir -p $O; export TMPDIR=/tmp
for knobs in "MN_X=0" "MN_WGRAD_STREAM=0" "MN_DETERMINISTIC=1" "MN_WGRAD_STREAM=0 MN_DETERMINISTIC=1" "MN_X=0"; do
  env $knobs timeout 300 python tools/second_step_probe.py fp32 1e-3 1.0 2>&1 | grep "^\["
done | tee $O/second_step_probe.txt
