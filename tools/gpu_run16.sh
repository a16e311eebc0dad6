#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== gpu tests"; timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/gpu_tests.log 2>&1; tail -4 gpurun_out/gpu_tests.log
TAG=v4 bash tools/gpu_prof.sh
