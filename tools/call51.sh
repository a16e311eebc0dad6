cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c51; O=$GRAFT_REPO_ROOT/gpurun_out/c51
CS=0,8,16,32 STEPS=30 timeout 900 python tools/rccl_rehearsal.py fp16x2m > $O/rehearsal_final_tree.txt 2> $O/err.txt
cat $O/rehearsal_final_tree.txt | cut -c1-200
