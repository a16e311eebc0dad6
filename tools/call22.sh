cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c22
BN_PROBE_POOL=1 timeout 600 tools/probes/bn_probe > gpurun_out/c22/pool_probe.txt 2>&1
cat gpurun_out/c22/pool_probe.txt
