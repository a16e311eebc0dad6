#!/bin/bash
# A/B of environment knobs on the bench step: [DT=fp16x2] tools/ab.sh "VAR=a VAR2=b" "VAR=c" ...   (each arm run twice, interleaved)
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
  for arm in "$@"; do
    v=$(env $arm python bench.py --dtype ${DT:-fp16x2m} --no-cpu-baseline --no-events --no-fast-mode --no-eval-metric --no-feed --steps ${STEPS:-30} --warmup 8 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "[${DT:-fp16x2m} $arm] $v"
  done
done
