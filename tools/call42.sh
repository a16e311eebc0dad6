cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c42; O=$GRAFT_REPO_ROOT/gpurun_out/c42
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tee $O/gpu_suite_summary.txt
DT=fp16 bash tools/ab.sh "MN_FWD_DS_SIDE=1" "MN_FWD_DS_SIDE=0" 2>&1 | tee $O/ab_fp16.txt
