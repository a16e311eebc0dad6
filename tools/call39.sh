cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c39; O=$GRAFT_REPO_ROOT/gpurun_out/c39
bash tools/ab.sh "MN_WGRAD_DEFER_STAGES=0" "MN_WGRAD_DEFER_STAGES=2" 2>&1 | tee $O/ab.txt
bash tools/ab.sh "MN_WGRAD_DEFER_STAGES=0" "MN_WGRAD_DEFER_STAGES=2" 2>&1 | tee -a $O/ab.txt
