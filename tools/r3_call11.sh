#!/bin/bash
# round 3, GPU call 11: igemm_halo register epilogue (MN_HALO_EPI=1) vs LDS-staged
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c11; mkdir -p $O
MN_HALO_EPI=1 MN_IGEMM_CONFIG=12 MN_IGEMM_HALO=1 timeout 300 python tests/forced_config_cases.py hip 2>&1 | tail -2 | tee $O/forced_256.txt
MN_HALO_EPI=1 MN_IGEMM_HALO=2 timeout 300 python tests/forced_config_cases.py hip 2>&1 | tail -2 | tee $O/forced_128.txt
for e in 0 1 0 1; do
  echo "== MN_HALO_EPI=$e" >> $O/halo_epi.txt
  MN_HALO_EPI=$e timeout 200 python tools/conv_bench.py fp16 2>&1 | grep -E "^layer(2|3|4) 3x3 (128|256|512)" | cut -c1-150 >> $O/halo_epi.txt
done
cat $O/halo_epi.txt
for rep in 1 2; do for e in 0 1; do
  MN_HALO_EPI=$e timeout 300 python bench.py --steps 50 --repeats 3 --no-cpu-baseline --no-parity-mode --no-events > $O/bench.json 2>> $O/bench.err
  python3 -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('epi $e', d['value'], d['ms_per_step'], d['config']['region_ms_per_step'])" | tee -a $O/bench_epi.txt
done; done
MN_HALO_EPI=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "full_size or train_step" 2>&1 | tail -3 | tee $O/pytest_epi.txt
