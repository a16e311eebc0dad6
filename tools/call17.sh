cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c17
for it in 1000 4000 16000; do BN_PROBE_HEAT=$it timeout 300 tools/probes/bn_probe > gpurun_out/c17/bn_probe_heat_$it.txt 2>&1; done
cat gpurun_out/c17/bn_probe_heat_*.txt
