cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c49; O=$GRAFT_REPO_ROOT/gpurun_out/c49
bash tools/ab.sh "MN_GATE_REC=1" "MN_GATE_REC=0" 2>&1 | tee $O/ab.txt
bash tools/ab.sh "MN_GATE_REC=1" "MN_GATE_REC=0" 2>&1 | tee -a $O/ab.txt
DT=fp16 bash tools/ab.sh "MN_GATE_REC=1" "MN_GATE_REC=0" 2>&1 | tee -a $O/ab.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "schedule or staged or fp16x2m or parity_mode" 2>&1 | grep -E "passed|failed|error" | tee $O/parity.txt
