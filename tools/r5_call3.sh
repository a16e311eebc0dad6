#!/bin/bash
# round 5, GPU call 3: fp16x2m with the stem's fp16 backward kernels (tests, A/B, profile); consumer-side BatchNorm fusion per launch;
# serial profile of the fp16 step by grid (bn_apply times per tensor size)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "fp16x2m or dropout" 2>&1 | tail -5 | tee $O/tests.txt
run() { env $2 python bench.py --dtype $1 --no-cpu-baseline --no-events --no-parity-mode --no-eval-metric --steps 30 --warmup 8 --repeats 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  for arm in "MN_STEM_BWD=1" "MN_STEM_BWD=0"; do
    echo "[fp16x2m $arm] $(run fp16x2m "$arm")" | tee -a $O/stem_bwd_ab.txt
  done
done
python tools/fbn_bench.py 2>&1 | tee $O/fbn_bench.txt
DT=fp16x2m TAG=c3 bash tools/prof_mode.sh
DT=fp16 TAG=c3 bash tools/prof_mode.sh
