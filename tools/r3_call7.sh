#!/bin/bash
# round 3, GPU call 7: software-pipelined fp32x3 K loop (operand splits under the previous group's MFMAs)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv_forward or conv_data_gradient or fp32x3 or adjoint" 2>&1 | tail -4 > $O/pytest.txt; cat $O/pytest.txt
timeout 300 python tools/conv_bench.py fp32x3 2>&1 | grep -E "^layer|plain GEMM" | cut -c1-170 > $O/conv_bench_x3.txt; cat $O/conv_bench_x3.txt
timeout 300 python bench.py --dtype fp32x3 --steps 30 --repeats 2 --no-cpu-baseline > $O/bench_x3.json 2>> $O/bench.err
python3 -c "import json;d=json.loads(open('$O/bench_x3.json').read().strip().splitlines()[-1]);print('x3', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])"
