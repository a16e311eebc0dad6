cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c47; O=$GRAFT_REPO_ROOT/gpurun_out/c47
bash tools/ab.sh "MN_X=0" "MN_HALO384=0" "MN_HALO384=2" "MN_H2_HALO384=0" "MN_H2_HALO384=2" "MN_BN_REDUCE_WGS=768" "MN_BN_REDUCE_WGS=384" "MN_HALO_A1=0" 2>&1 | tee $O/ab.txt
