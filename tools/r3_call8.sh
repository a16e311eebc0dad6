#!/bin/bash
# round 3, GPU call 8: tap-split fused weight gradient (MN_WGF_TS=1: a B fragment feeds two MFMAs) vs the round-2 form
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c8; mkdir -p $O
MN_WGF_TS=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "weight_gradient or adjoint or race_screen" 2>&1 | tail -4 > $O/pytest_ts.txt; cat $O/pytest_ts.txt
for ts in 0 1 0 1; do
  echo "== MN_WGF_TS=$ts" >> $O/conv_bench_wgf_ts.txt
  MN_WGF_TS=$ts timeout 300 python tools/conv_bench.py fp16 2>&1 | grep -E "wgrad through the workspace" | cut -c1-150 >> $O/conv_bench_wgf_ts.txt
done
cat $O/conv_bench_wgf_ts.txt
for rep in 1 2; do for ts in 0 1; do
  MN_WGF_TS=$ts timeout 300 python bench.py --steps 50 --repeats 3 --no-cpu-baseline --no-parity-mode --no-events > $O/bench.json 2>> $O/bench.err
  python3 -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('ts $ts', d['value'], d['ms_per_step'], d['config']['region_ms_per_step'])" | tee -a $O/bench_ts.txt
done; done
