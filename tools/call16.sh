cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c16
timeout 600 tools/probes/bn_probe > gpurun_out/c16/bn_probe.txt 2>&1
BN_PROBE_GRID=16384 timeout 600 tools/probes/bn_probe > gpurun_out/c16/bn_probe_grid16384.txt 2>&1
tail -30 gpurun_out/c16/bn_probe.txt
