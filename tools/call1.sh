cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c1; O=gpurun_out/c1
rocminfo | grep -m3 -E "Marketing|Compute Unit" > $O/box.txt; nproc >> $O/box.txt
python bench.py --no-cpu-baseline --no-eval-metric --steps 50 --repeats 3 > $O/bench_head.json 2> $O/bench_head.err
for m in churn keep pool; do timeout 600 python tools/arena_probe.py $m > $O/arena_$m.txt 2>&1; done
DT=fp16x2m MODELS=4 timeout 600 python tools/arena_probe.py churn > $O/arena_churn_fp16x2m.txt 2>&1
TAG=c1 bash tools/sq_counters.sh > $O/sq.log 2>&1
