#!/bin/bash
# round 3, GPU call 1: fp32x3 (split-operand) mode -- hardware probe, parity, per-layer timing, bench in all three modes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c1; mkdir -p $O
./tools/probes/mfma_denorm_probe > $O/denorm_probe.txt 2>&1; cat $O/denorm_probe.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv_forward or conv_data_gradient or conv_weight_gradient or stem_conv or fp32x3" 2>&1 | tail -15 > $O/pytest_x3.txt; cat $O/pytest_x3.txt
timeout 600 python tools/conv_bench.py fp32x3 > $O/conv_bench_fp32x3.txt 2>&1; cat $O/conv_bench_fp32x3.txt
timeout 300 python tools/conv_bench.py fp32 > $O/conv_bench_fp32.txt 2>&1; tail -12 $O/conv_bench_fp32.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json
timeout 300 python bench.py --dtype fp32 --steps 20 --repeats 2 --no-cpu-baseline > $O/bench_fp32.json 2>> $O/bench.err; tail -1 $O/bench_fp32.json
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "full_size_parity_configs2 or fp16_close" 2>&1 | tail -8 > $O/pytest_full.txt; cat $O/pytest_full.txt
