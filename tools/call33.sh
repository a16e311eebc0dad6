cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c33; O=$GRAFT_REPO_ROOT/gpurun_out/c33
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tee $O/gpu_suite_summary.txt
