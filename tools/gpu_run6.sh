#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== conv bench"; timeout 300 python tools/conv_bench.py fp16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/conv_bench.log
echo "== gpu tests"; timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/gpu_tests.log 2>&1; tail -4 gpurun_out/gpu_tests.log
TAG=v3 bash tools/gpu_prof.sh
