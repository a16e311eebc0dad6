#!/bin/bash
# End-of-round check on the GPU box (gpurun -- 'bash tools/final_check.sh'): the default bench line, serial kernel stats of the
# fp16 step, the full GPU suite and smoke(); everything lands in gpurun_out/final_$TAG.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_${TAG:-cur}; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench.err; tail -1 $O/bench_default.json | cut -c1-300
cd /tmp && MN_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_serial -o r -- python $R/bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-events --no-parity-mode > $O/rocprof.log 2>&1
cp /tmp/prof_serial/r_kernel_stats.csv $O/kernel_stats_serial.csv
cd $R && timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tee $O/gpu_suite_summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -7 | tee $O/smoke.txt
