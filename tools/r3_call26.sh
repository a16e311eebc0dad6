#!/bin/bash
# round 3, GPU call 26: fp32x3 fused weight gradient with software-pipelined fragment reads (prev = reads and MFMAs per item in order)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c26; mkdir -p $O
PREV=$PWD/tools/ablation/libmapnet_hip_prev.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "weight_gradient or through_workspace" 2>&1 | tail -2 | tee $O/pytest.txt
for l in prev new prev new; do
  if [ $l = prev ]; then export MN_LIB=$PREV; else unset MN_LIB; fi
  echo "== $l" >> $O/wgrad_x3.txt
  timeout 300 python tools/conv_bench.py fp32x3 2>&1 | grep -E "wgrad through the workspace" | cut -c1-100 >> $O/wgrad_x3.txt
done
unset MN_LIB; cat $O/wgrad_x3.txt
for rep in 1 2; do for l in prev new; do
  if [ $l = prev ]; then export MN_LIB=$PREV; else unset MN_LIB; fi
  timeout 300 python bench.py --dtype fp32x3 --steps 30 --repeats 3 --no-cpu-baseline --no-parity-mode --no-events > $O/bench.json 2>> $O/bench.err
  python3 -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('$l fp32x3', d['value'], d['ms_per_step'], d['config']['region_ms_per_step'])" | tee -a $O/bench_x3.txt
done; done
