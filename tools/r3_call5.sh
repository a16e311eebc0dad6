#!/bin/bash
# round 3, GPU call 5: whole GPU suite after the pruning; main-loop ablation of the fp32x3 implicit-GEMM kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c5; mkdir -p $O
ABL=$PWD/tools/ablation/libmapnet_hip_abl.so
for a in 0 1 2 4 8 16 24 3 7 31; do
  echo "== MN_ABLATE=$a (1 no DMA, 2 no fragment reads, 4 no barrier, 8 no operand splits, 16 no MFMAs)" >> $O/ablation_x3.txt
  MN_ABLATE=$a MN_LIB=$ABL CB_MATCH="layer3" timeout 200 python tools/conv_bench.py fp32x3 2>&1 | grep -E "^layer3" | cut -c1-140 >> $O/ablation_x3.txt
done
cat $O/ablation_x3.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 600 python bench.py --no-cpu-baseline --repeats 3 > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp16', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'], '| x3', d['parity_mode']['value'], d['parity_mode']['ms_per_step'], d['parity_mode']['roofline']['conv_ms_per_step'])"
