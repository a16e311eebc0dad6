cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c41; O=$GRAFT_REPO_ROOT/gpurun_out/c41
bash tools/ab.sh "MN_FWD_DS_SIDE=0" "MN_FWD_DS_SIDE=1" 2>&1 | tee $O/ab.txt
bash tools/ab.sh "MN_FWD_DS_SIDE=0" "MN_FWD_DS_SIDE=1" 2>&1 | tee -a $O/ab.txt
