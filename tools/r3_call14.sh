#!/bin/bash
# round 3, GPU call 14: fused weight gradient, DMA issue position inside the step (-1 after the barrier; after item 1 / 4 / 7 of 10)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c14; mkdir -p $O
for d in 1 4 7; do MN_WGF_DMAPOS=$d timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "weight_gradient" 2>&1 | tail -1; done
for e in -1 1 4 7 -1 1 4 7; do
  echo "== MN_WGF_DMAPOS=$e" >> $O/wgf_dmapos.txt
  MN_WGF_DMAPOS=$e timeout 200 python tools/conv_bench.py fp16 2>&1 | grep -E "wgrad through the workspace" | cut -c1-120 >> $O/wgf_dmapos.txt
done
cat $O/wgf_dmapos.txt
for rep in 1 2; do for e in -1 1 4 7; do
  MN_WGF_DMAPOS=$e timeout 300 python bench.py --steps 50 --repeats 3 --no-cpu-baseline --no-parity-mode --no-events > $O/bench.json 2>> $O/bench.err
  python3 -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('dmapos $e', d['value'], d['ms_per_step'], d['config']['region_ms_per_step'])" | tee -a $O/bench_dmapos.txt
done; done
