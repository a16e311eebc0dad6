cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c6; O=$GRAFT_REPO_ROOT/gpurun_out/c6
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "schedule or fp16_close or eval_forward or eval_flow" > $O/tests_fp16_gates.txt 2>&1; tail -3 $O/tests_fp16_gates.txt
DT=fp16x2m STEPS=40 bash tools/ab.sh "MN_WGRAD_SCHED=0" "MN_WGRAD_SCHED=1" "MN_WGRAD_SCHED=2" "MN_BN_REDUCE_WGS=1024" "MN_BN_REDUCE_WGS=256" > $O/ab_sched_fp16x2m.txt 2>&1
cat $O/ab_sched_fp16x2m.txt
