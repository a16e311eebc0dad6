#!/bin/bash
# round 3, GPU call 15: fused weight gradient DMA position 7 / 8 / 9; igemm.h with the DMA issue behind the first MFMA group
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c15; mkdir -p $O
DL=$PWD/tools/ablation/libmapnet_hip_dlate.so
for d in 8 9; do MN_WGF_DMAPOS=$d timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "weight_gradient" 2>&1 | tail -1; done
MN_LIB=$DL timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv_forward or conv_data_gradient or full_size" 2>&1 | tail -1 | tee $O/pytest_dlate.txt
for e in 7 8 9 7 8 9; do
  echo "== MN_WGF_DMAPOS=$e" >> $O/wgf_dmapos.txt
  MN_WGF_DMAPOS=$e timeout 200 python tools/conv_bench.py fp16 2>&1 | grep -E "wgrad through the workspace" | cut -c1-120 >> $O/wgf_dmapos.txt
done
cat $O/wgf_dmapos.txt
for l in main dlate main dlate; do
  echo "== $l fp16" >> $O/dlate.txt
  if [ $l = dlate ]; then export MN_LIB=$DL; else unset MN_LIB; fi
  timeout 200 python tools/conv_bench.py fp16 2>&1 | grep -E "^(layer|down|stem)" | grep -v "through the workspace" | cut -c1-130 >> $O/dlate.txt
  echo "== $l fp32x3" >> $O/dlate.txt
  timeout 300 python tools/conv_bench.py fp32x3 2>&1 | grep -E "^(layer|down|stem)" | grep -v "through the workspace" | cut -c1-130 >> $O/dlate.txt
done
unset MN_LIB
cat $O/dlate.txt
for rep in 1 2; do for l in main dlate; do
  if [ $l = dlate ]; then export MN_LIB=$DL; else unset MN_LIB; fi
  timeout 300 python bench.py --steps 50 --repeats 3 --no-cpu-baseline --no-parity-mode --no-events > $O/bench.json 2>> $O/bench.err
  python3 -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('$l fp16', d['value'], d['ms_per_step'], d['config']['region_ms_per_step'])" | tee -a $O/bench_dlate.txt
  timeout 300 python bench.py --dtype fp32x3 --steps 20 --repeats 3 --no-cpu-baseline --no-parity-mode --no-events > $O/bench.json 2>> $O/bench.err
  python3 -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('$l fp32x3', d['value'], d['ms_per_step'], d['config']['region_ms_per_step'])" | tee -a $O/bench_dlate.txt
done; done
unset MN_LIB
for rep in 1 2; do for e in 7 8 9; do
  MN_WGF_DMAPOS=$e timeout 300 python bench.py --steps 50 --repeats 3 --no-cpu-baseline --no-parity-mode --no-events > $O/bench.json 2>> $O/bench.err
  python3 -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('wgf dmapos $e', d['value'], d['ms_per_step'], d['config']['region_ms_per_step'])" | tee -a $O/bench_wgf.txt
done; done
