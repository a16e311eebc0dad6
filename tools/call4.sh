cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c4; O=$GRAFT_REPO_ROOT/gpurun_out/c4
DT=fp16x2m STEPS=40 bash tools/ab.sh "MN_BN_REC=0" "MN_BN_REC=1" "MN_BN_REC=1 MN_BN_REDUCE_U=2" > $O/ab_record_fp16x2m.txt 2>&1
MN_RECORD_DEVIATIONS=$O/deviations.jsonl timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fp16x2m or configs2 or uint8 or eval_forward or eval_flow or stem or fp16_close or schedule" > $O/tests_mixed.txt 2>&1
DT=fp16x2m TAG=c4 bash tools/prof_mode.sh > $O/prof.log 2>&1
python bench.py --no-eval-metric --no-feed --no-fast-mode --steps 50 --repeats 3 > $O/bench.json 2> $O/bench.err
