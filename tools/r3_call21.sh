#!/bin/bash
# round 3, GPU call 21: fp32x3 tap-fused weight gradient (MN_WGRAD_FUSED=0 = the plain-GEMM x3 kernel)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c21; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "weight_gradient or through_workspace or fp32x3" 2>&1 | tail -3 | tee $O/pytest.txt
for f in 0 1 0 1; do
  echo "== MN_WGRAD_FUSED=$f fp32x3" >> $O/wgrad_x3.txt
  MN_WGRAD_FUSED=$f timeout 300 python tools/conv_bench.py fp32x3 2>&1 | grep -E "^layer[1-4] 3x3 (64|128|256|512)" | cut -c1-200 >> $O/wgrad_x3.txt
done
cat $O/wgrad_x3.txt
for rep in 1 2; do for f in 0 1; do
  MN_WGRAD_FUSED=$f timeout 300 python bench.py --dtype fp32x3 --steps 30 --repeats 3 --no-cpu-baseline --no-parity-mode --no-events > $O/bench.json 2>> $O/bench.err
  python3 -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('fused $f fp32x3', d['value'], d['ms_per_step'], d['config']['region_ms_per_step'])" | tee -a $O/bench_x3.txt
done; done
