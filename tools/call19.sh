cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c19
timeout 600 tools/probes/bn_probe > gpurun_out/c19/bn_probe.txt 2>&1
grep -E "layer|channels/thread|bwd_apply_rec wgs" gpurun_out/c19/bn_probe.txt
