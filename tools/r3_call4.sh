#!/bin/bash
# round 3, GPU call 4: BatchNorm coefficient tables in LDS ([e][piece], conflict-free) -> registers
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c4; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "batchnorm or maxpool or train_step_fp32_parity_small or deterministic_mode_is" 2>&1 | tail -4 > $O/pytest.txt; cat $O/pytest.txt
timeout 900 python bench.py --no-cpu-baseline --repeats 3 > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp16', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'], '| x3', d['parity_mode']['value'], d['parity_mode']['ms_per_step'], d['parity_mode']['roofline']['conv_ms_per_step'])"
R=$GRAFT_REPO_ROOT
cd /tmp && MN_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_serial -o r -- python $R/bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-events --no-parity-mode > $R/$O/rocprof.log 2>&1
cp /tmp/prof_serial/r_kernel_stats.csv $R/$O/kernel_stats_serial_fp16.csv
head -12 $R/$O/kernel_stats_serial_fp16.csv | cut -c1-150
