cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c15
timeout 600 tools/probes/stream_probe > gpurun_out/c15/stream_probe_layer1.txt 2>&1
timeout 300 tools/probes/stream_probe $((192*28*28*128)) > gpurun_out/c15/stream_probe_layer2.txt 2>&1
timeout 300 tools/probes/stream_probe $((192*14*14*256)) > gpurun_out/c15/stream_probe_layer3.txt 2>&1
tail -5 gpurun_out/c15/stream_probe_layer1.txt
