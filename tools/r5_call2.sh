#!/bin/bash
# round 5, GPU call 2: new tests, serial profile of fp16x2m, CU-mask partition of the weight-gradient stream,
# weight-gradient schedule of fp16x2m, accuracy of fp16 / fp16x2m / fp16x2 over five seeds under MN_DETERMINISTIC=1 (item 5)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "overflow or deterministic_mode_is_bit_reproducible" 2>&1 | tail -5 | tee $O/tests.txt
DT=fp16x2m TAG=c2 bash tools/prof_mode.sh
run() { env $2 python bench.py --dtype $1 --no-cpu-baseline --no-events --no-parity-mode --no-eval-metric --steps 30 --warmup 8 --repeats 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  for arm in "X=0" "MN_WGRAD_CUS=64" "MN_WGRAD_CUS=128" "MN_WGRAD_CUS=128 MN_WGRAD_CU_SPREAD=1" "MN_WGRAD_CUS=192" "MN_WGRAD_CUS=224"; do
    echo "[fp16 $arm] $(run fp16 "$arm")" | tee -a $O/cu_mask.txt
  done
  for arm in "X=0" "MN_WGRAD_CUS=128" "MN_WGRAD_CUS=192" "MN_WGRAD_SCHED=1" "MN_WGRAD_SCHED=2"; do
    echo "[fp16x2m $arm] $(run fp16x2m "$arm")" | tee -a $O/cu_mask.txt
  done
done
timeout 1500 python tools/accuracy_eval.py --dtypes fp16,fp16x2m,fp16x2 --seeds 7,8,9,10,11 --deterministic > $O/accuracy_deterministic_five_seeds.json 2> $O/accuracy_deterministic_five_seeds.log
grep "^#" $O/accuracy_deterministic_five_seeds.log
