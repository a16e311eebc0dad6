cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c5; O=$GRAFT_REPO_ROOT/gpurun_out/c5
MN_RECORD_DEVIATIONS=$O/deviations.jsonl timeout 2700 python -m pytest tests -m gpu -q > $O/gpu_suite.txt 2>&1
tail -5 $O/gpu_suite.txt
