#!/bin/bash
# round 5, GPU call 1: the fp16x2m mode -- its GPU tests, whole-step time beside fp16x2 and fp16, serial kernel profile
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "fp16x2m" 2>&1 | tail -5 | tee $O/tests_fp16x2m.txt
for rep in 1 2; do for dt in fp16x2m fp16x2 fp16; do
  v=$(python bench.py --dtype $dt --no-cpu-baseline --no-events --no-parity-mode --no-eval-metric --steps 30 --warmup 8 --repeats 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "[$dt] $v" | tee -a $O/whole_step.txt
done; done
DT=fp16x2m TAG=c1 bash tools/prof_mode.sh
