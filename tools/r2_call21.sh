#!/bin/bash
# Round-2 GPU call 21/22: the second / third step of the HIP path against the oracle in fp32 AND in fp64
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c22; mkdir -p $O; export TMPDIR=/tmp
( timeout 600 python tools/second_step_probe.py fp32 1e-3 1.0 3; MN_DETERMINISTIC=1 timeout 600 python tools/second_step_probe.py fp32 1e-3 1.0 3; timeout 600 python tools/second_step_probe.py fp32 1e-4 1e-8 3 ) 2>&1 | grep -E "^\[|^step" | tee $O/second_step_probe.txt
