#!/bin/bash
# round 3, GPU call 10: fused weight gradient with 4 DMA steps in flight; prologue / K-loop / epilogue split of igemm_halo
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c10; mkdir -p $O
ABL=$PWD/tools/ablation/libmapnet_hip_abl.so
MN_WGF_DEPTH=4 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "weight_gradient" 2>&1 | tail -3 | tee $O/pytest_d4.txt
for d in 3 4 3 4; do
  echo "== MN_WGF_DEPTH=$d" >> $O/wgf_depth.txt
  MN_WGF_DEPTH=$d timeout 300 python tools/conv_bench.py fp16 2>&1 | grep -E "wgrad through the workspace" | cut -c1-150 >> $O/wgf_depth.txt
done
cat $O/wgf_depth.txt
for a in 0 1 2 3; do
  echo "== MN_HALO_ABLATE=$a (1 no epilogue, 2 one K-step only, 3 both = prologue only)" >> $O/igemm_halo_ablation.txt
  MN_HALO_ABLATE=$a MN_LIB=$ABL timeout 200 python tools/conv_bench.py fp16 2>&1 | grep -E "^layer(2|3|4) 3x3 (128|256|512)" | cut -c1-150 >> $O/igemm_halo_ablation.txt
done
cat $O/igemm_halo_ablation.txt
for rep in 1 2; do for d in 3 4; do
  MN_WGF_DEPTH=$d timeout 300 python bench.py --steps 50 --repeats 3 --no-cpu-baseline --no-parity-mode --no-events > $O/bench.json 2>> $O/bench.err
  python3 -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('depth $d', d['value'], d['ms_per_step'], d['config']['region_ms_per_step'])" | tee -a $O/bench_depth.txt
done; done
