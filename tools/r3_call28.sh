#!/bin/bash
# round 3, GPU call 28: stem BatchNorm-backward sums in pooled-window order (prev = pixel order)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c28; mkdir -p $O
PREV=$PWD/tools/ablation/libmapnet_hip_prev.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "stem_backward or train_step or full_size_parity_mapnet" 2>&1 | tail -2 | tee $O/pytest.txt
for l in prev new prev new; do
  if [ $l = prev ]; then export MN_LIB=$PREV; else unset MN_LIB; fi
  echo "== $l" >> $O/stem_bwd.txt
  timeout 300 python tools/conv_bench.py fp16 2>&1 | grep -E "^stem" | cut -c1-140 >> $O/stem_bwd.txt
done
unset MN_LIB; cat $O/stem_bwd.txt
for rep in 1 2 3; do for l in prev new; do
  if [ $l = prev ]; then export MN_LIB=$PREV; else unset MN_LIB; fi
  timeout 300 python bench.py --steps 50 --repeats 3 --no-cpu-baseline --no-parity-mode --no-events > $O/bench.json 2>> $O/bench.err
  python3 -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('$l fp16', d['value'], d['ms_per_step'], d['config']['region_ms_per_step'])" | tee -a $O/bench.txt
done; done
