cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c43; O=$GRAFT_REPO_ROOT/gpurun_out/c43
bash tools/ab.sh "MN_HEAD_WGRAD_SIDE=1" "MN_HEAD_WGRAD_SIDE=0" 2>&1 | tee $O/ab.txt
bash tools/ab.sh "MN_HEAD_WGRAD_SIDE=1" "MN_HEAD_WGRAD_SIDE=0" 2>&1 | tee -a $O/ab.txt
