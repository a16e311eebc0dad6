#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== gpu tests"; timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/gpu_tests.log 2>&1; tail -4 gpurun_out/gpu_tests.log
echo "== bench graphs on"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1
echo "== bench graphs off"; MN_GRAPHS=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1
