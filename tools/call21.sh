cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c21; O=gpurun_out/c21
bash tools/ab.sh "MN_X=0" "MN_HALO_A1_F16=2" "MN_HALO_A1_F16=0" "MN_HALO_A1_F16=2 MN_WGRAD_SCHED=0" 2>&1 | tee $O/ab.txt
