#!/bin/bash
# round 3, GPU call 9: rotated loop of the fused weight gradient (next step's first fragments requested across the barrier)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "weight_gradient or adjoint or race_screen or deterministic_mode_is" 2>&1 | tail -4 > $O/pytest.txt; cat $O/pytest.txt
timeout 300 python tools/conv_bench.py fp16 2>&1 | grep -E "wgrad through the workspace" | cut -c1-150 | tee $O/conv_bench_wgf.txt
for rep in 1 2; do
  timeout 300 python bench.py --steps 50 --repeats 3 --no-cpu-baseline --no-parity-mode --no-events > $O/bench.json 2>> $O/bench.err
  python3 -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('rotated', d['value'], d['ms_per_step'], d['config']['region_ms_per_step'])" | tee -a $O/bench.txt
done
