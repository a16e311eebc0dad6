cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c44; O=$GRAFT_REPO_ROOT/gpurun_out/c44
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tee $O/gpu_suite_summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -9 | tee $O/smoke.txt
