#!/bin/bash
# round 3, GPU call 6: ping-pong form of the fp32x3 implicit-GEMM kernel (MN_X3_PP=1) vs two independent workgroups per CU
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c6; mkdir -p $O
MN_X3_PP=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv_forward or conv_data_gradient or fp32x3" 2>&1 | tail -4 > $O/pytest_pp.txt; cat $O/pytest_pp.txt
for pp in 0 1; do
  echo "== MN_X3_PP=$pp" >> $O/conv_bench_x3_pp.txt
  MN_X3_PP=$pp timeout 300 python tools/conv_bench.py fp32x3 2>&1 | grep -E "^layer|plain GEMM" | cut -c1-150 >> $O/conv_bench_x3_pp.txt
done
cat $O/conv_bench_x3_pp.txt
for pp in 0 1; do
  MN_X3_PP=$pp timeout 300 python bench.py --dtype fp32x3 --steps 30 --repeats 2 --no-cpu-baseline > $O/bench_x3.json 2>> $O/bench.err
  python3 -c "import json;d=json.loads(open('$O/bench_x3.json').read().strip().splitlines()[-1]);print('pp $pp', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])" | tee -a $O/bench_x3_pp.txt
done
