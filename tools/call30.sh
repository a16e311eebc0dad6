cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c30; O=$GRAFT_REPO_ROOT/gpurun_out/c30
bash tools/ab.sh "MN_WGRAD_TAIL=0" "MN_WGRAD_TAIL=1" 2>&1 | tee $O/ab.txt
bash tools/ab.sh "MN_WGRAD_TAIL=0" "MN_WGRAD_TAIL=1" 2>&1 | tee -a $O/ab.txt
DT=fp16 bash tools/ab.sh "MN_WGRAD_TAIL=0" "MN_WGRAD_TAIL=1" "MN_WGRAD_TAIL=2" 2>&1 | tee -a $O/ab.txt
