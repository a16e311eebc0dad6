#!/bin/bash
# Round-2 GPU call 24: full GPU suite + bench line on the final committed tree
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c30; mkdir -p $O; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/gpu_suite.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/gpu_suite.log | tail -8
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --no-cpu-baseline > $O/bench_fp16.json 2> $O/bench_fp16.err; tail -1 $O/bench_fp16.json | cut -c1-400
