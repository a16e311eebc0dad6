cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c32; O=$GRAFT_REPO_ROOT/gpurun_out/c32
DT=fp16 bash tools/ab.sh "MN_WGRAD_TAIL=0" "MN_WGRAD_TAIL=1" "MN_WGRAD_TAIL=3 MN_STEM_WGRAD_PER_CU=1" "MN_WGRAD_TAIL=2 MN_STEM_WGRAD_PER_CU=1" 2>&1 | tee $O/ab.txt
bash tools/ab.sh "MN_WGRAD_TAIL=0" "MN_WGRAD_TAIL=1" "MN_WGRAD_TAIL=3 MN_STEM_WGRAD_PER_CU=1" 2>&1 | tee -a $O/ab.txt
