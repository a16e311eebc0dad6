cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c54; O=$GRAFT_REPO_ROOT/gpurun_out/c54
DT=fp16 bash tools/ab.sh "MN_X=0" "MN_WGRAD_SCHED=0 MN_WGRAD_EARLY_STAGES=15 MN_WGRAD_DEFER_STAGES=0" 2>&1 | tee $O/ab.txt
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tee $O/gpu_suite_summary.txt
