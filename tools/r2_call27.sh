#!/bin/bash
# Round-2 GPU call 27: SQ counters (LDS bank conflicts, MFMA busy, waits) of the layer1 / layer2 kernels for the next round
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -f gpurun_out/pmc_conv_l1.txt gpurun_out/pmc_conv_l2.txt
bash tools/pmc_conv.sh "layer1" l1
bash tools/pmc_conv.sh "layer2 3x3" l2
wc -l gpurun_out/pmc_conv_l1.txt gpurun_out/pmc_conv_l2.txt
