#!/bin/bash
# round 3, GPU call 2: fp32x3 variants -- conversions (portable vs inline asm), tile shapes, the new weight-gradient kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c2; mkdir -p $O
X3ASM=$PWD/tools/ablation/libmapnet_hip_x3asm.so
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "conv_weight_gradient or stem_conv" 2>&1 | tail -4 > $O/pytest_wgrad.txt; cat $O/pytest_wgrad.txt
for cfg in 0 1 2; do
  echo "== portable conversions, MN_X3_CFG=$cfg" >> $O/conv_bench_x3_variants.txt
  MN_X3_CFG=$cfg timeout 300 python tools/conv_bench.py fp32x3 2>&1 | grep -E "^layer|^stem|plain GEMM M=67584" >> $O/conv_bench_x3_variants.txt
  echo "== inline-asm conversions, MN_X3_CFG=$cfg" >> $O/conv_bench_x3_variants.txt
  MN_X3_CFG=$cfg MN_LIB=$X3ASM timeout 300 python tools/conv_bench.py fp32x3 2>&1 | grep -E "^layer|^stem|plain GEMM M=67584" >> $O/conv_bench_x3_variants.txt
done
cat $O/conv_bench_x3_variants.txt
for cfg in 0 2; do for lib in "" $X3ASM; do
  MN_LIB=$lib MN_X3_CFG=$cfg timeout 300 python bench.py --dtype fp32x3 --steps 30 --repeats 2 --no-cpu-baseline > $O/bench_x3.json 2>> $O/bench.err
  python3 -c "import json;d=json.loads(open('$O/bench_x3.json').read().strip().splitlines()[-1]);print('cfg $cfg lib [$lib]', d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])" | tee -a $O/bench_x3_variants.txt
done; done
