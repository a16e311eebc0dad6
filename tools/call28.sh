cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c28; O=$GRAFT_REPO_ROOT/gpurun_out/c28
bash tools/ab.sh "MN_X=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" 2>&1 | tee $O/ab.txt
