cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c20; O=gpurun_out/c20
bash tools/ab.sh "MN_X=0" "MN_BN_APPLY_V4=0 MN_BN_BWD_APPLY_FULL=999999999" "MN_BN_BWD_APPLY_FULL=999999999" "MN_BN_APPLY_WGS=16384" "MN_EW_WGS=1000000" 2>&1 | tee $O/ab.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "h2 or fp16x2 or step or train" 2>&1 | tail -3 | tee $O/parity.txt
