cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c46; O=$GRAFT_REPO_ROOT/gpurun_out/c46
bash tools/ab.sh "MN_BWD_DS_SIDE=0" "MN_BWD_DS_SIDE=1" 2>&1 | tee $O/ab.txt
bash tools/ab.sh "MN_BWD_DS_SIDE=0" "MN_BWD_DS_SIDE=1" 2>&1 | tee -a $O/ab.txt
timeout 900 env MN_BWD_DS_SIDE=1 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "schedule or full_resolution or staged" 2>&1 | grep -E "passed|failed|error" | tee $O/parity.txt
