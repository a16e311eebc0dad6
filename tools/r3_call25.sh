#!/bin/bash
# round 3, GPU call 25: ablation of the fp32x3 fused weight gradient (MN_WGF_ABLATE: 1 no loads/splits/stores in the loop,
# 2 fragment reads only in the first step, 4 no MFMAs, 8 no barrier per step) and of the fp16 kernel (1 no DMA, 2 no B reads, 4 no MFMA)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c25; mkdir -p $O
ABL=$PWD/tools/ablation/libmapnet_hip_abl.so
for a in 0 1 2 4 8 3 5 6 7 15; do
  echo "== fp32x3 MN_WGF_ABLATE=$a" >> $O/wgf_x3_ablation.txt
  MN_WGF_ABLATE=$a MN_LIB=$ABL timeout 300 python tools/conv_bench.py fp32x3 2>&1 | grep -E "wgrad through the workspace" | cut -c1-100 >> $O/wgf_x3_ablation.txt
done
cat $O/wgf_x3_ablation.txt
for a in 0 1 2 4; do
  echo "== fp16 MN_WGF_ABLATE=$a" >> $O/wgf_fp16_ablation.txt
  MN_WGF_ABLATE=$a MN_LIB=$ABL timeout 300 python tools/conv_bench.py fp16 2>&1 | grep -E "wgrad through the workspace" | cut -c1-100 >> $O/wgf_fp16_ablation.txt
done
cat $O/wgf_fp16_ablation.txt
