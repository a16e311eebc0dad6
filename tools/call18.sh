cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c18
timeout 600 tools/probes/bn_probe > gpurun_out/c18/bn_probe.txt 2>&1
BN_PROBE_HEAT=4000 timeout 300 tools/probes/bn_probe > gpurun_out/c18/bn_probe_heat.txt 2>&1
timeout 600 tools/probes/stream_probe > gpurun_out/c18/stream_probe_layer1.txt 2>&1
timeout 300 tools/probes/stream_probe $((192*16*22*256)) > gpurun_out/c18/stream_probe_layer3.txt 2>&1
head -16 gpurun_out/c18/bn_probe.txt
