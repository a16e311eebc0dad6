#!/bin/bash
# round 3, GPU call 3: register-resident BatchNorm coefficients, own fill kernel for the accumulators, asm conversions default
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c3; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "batchnorm or maxpool or train_step or deterministic_mode_is or full_size_parity_configs2 or stem" 2>&1 | tail -6 > $O/pytest.txt; cat $O/pytest.txt
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp16', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'], '| x3', d['parity_mode']['value'], d['parity_mode']['ms_per_step'], d['parity_mode']['roofline']['conv_ms_per_step'])"
R=$GRAFT_REPO_ROOT
cd /tmp && MN_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_serial -o r -- python $R/bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-events --no-parity-mode > $R/$O/rocprof.log 2>&1
cp /tmp/prof_serial/r_kernel_stats.csv $R/$O/kernel_stats_serial_fp16.csv
head -25 $R/$O/kernel_stats_serial_fp16.csv | cut -c1-170
cd /tmp && MN_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_serial_x3 -o r -- python $R/bench.py --dtype fp32x3 --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-events > $R/$O/rocprof_x3.log 2>&1
cp /tmp/prof_serial_x3/r_kernel_stats.csv $R/$O/kernel_stats_serial_fp32x3.csv
head -25 $R/$O/kernel_stats_serial_fp32x3.csv | cut -c1-170
