cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c3; O=$GRAFT_REPO_ROOT/gpurun_out/c3
DT=fp16x2m STEPS=40 bash tools/ab.sh "MN_LIB=$GRAFT_REPO_ROOT/tools/ablation/libmapnet_hip_before_copies.so" "MN_AFTER=1" > $O/ab_copies_fp16x2m.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fp16x2m or configs2 or configs1 or feed or stand_in or weight_grad or wgrad or dgrad or halo or schedule or deterministic" > $O/tests_mixed.txt 2>&1
LDS_KB=64 CS=0,8,16,32,64 timeout 900 python tools/rccl_rehearsal.py fp16x2m > $O/rccl_rehearsal_lds64.txt 2>&1
MN_RECORD_DEVIATIONS=$O/deviations.jsonl timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "train_step or eval_forward or eval_flow" > $O/tests_record.txt 2>&1
python bench.py --no-eval-metric --no-feed --steps 50 --repeats 3 > $O/bench.json 2> $O/bench.err
