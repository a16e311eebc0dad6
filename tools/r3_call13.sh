#!/bin/bash
# round 3, GPU call 13: igemm_halo DMA issue position inside the K-step (0 after the barrier, 1 / 2 after the first / second MFMA group)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c13; mkdir -p $O
MN_HALO_DMAPOS=1 MN_IGEMM_CONFIG=12 MN_IGEMM_HALO=1 timeout 300 python tests/forced_config_cases.py hip 2>&1 | tail -1
MN_HALO_DMAPOS=2 MN_IGEMM_HALO=2 timeout 300 python tests/forced_config_cases.py hip 2>&1 | tail -1
for e in 0 1 2 0 1 2; do
  echo "== MN_HALO_DMAPOS=$e" >> $O/halo_dmapos.txt
  MN_HALO_DMAPOS=$e timeout 200 python tools/conv_bench.py fp16 2>&1 | grep -E "^layer(2|3|4) 3x3 (128|256|512)" | grep -v "through the workspace" | cut -c1-120 >> $O/halo_dmapos.txt
done
cat $O/halo_dmapos.txt
