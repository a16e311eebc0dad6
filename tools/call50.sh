cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c50; O=$GRAFT_REPO_ROOT/gpurun_out/c50
timeout 1200 python bench.py --all-modes --steps 40 --warmup 8 --repeats 3 --no-eval-metric --no-feed > $O/bench_all_modes.json 2> $O/bench.err
python3 - <<PY
import json
d=json.loads(open('$O/bench_all_modes.json').read().strip().splitlines()[-1])
print('fp16x2m', d['value'], d['ms_per_step'], d.get('meets_tolerance'))
for k in ('fast_mode','experimental_mode','parity_mode_full'):
    if k in d: print(k, d[k]['dtype_mode'], d[k]['value'], d[k]['ms_per_step'], {kk:d[k].get('parity',{}).get(kk) for kk in ('loss_rel','pose_abs_max','grad_l2_rel_all')})
PY
for dt in fp32x3 fp32; do python bench.py --dtype $dt --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-events --no-fast-mode --no-eval-metric --no-feed 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$dt', d['value'], d['ms_per_step'])"; done
