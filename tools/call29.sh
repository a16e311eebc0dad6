cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c29; O=$GRAFT_REPO_ROOT/gpurun_out/c29
bash tools/ab.sh "MN_WGRAD_TAIL=0" "MN_WGRAD_TAIL=1" "MN_WGRAD_TAIL=2" "MN_WGRAD_TAIL=3" 2>&1 | tee $O/ab.txt
