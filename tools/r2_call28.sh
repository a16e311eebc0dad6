#!/bin/bash
# Round-2 GPU call 28: halo_pp with 8x32-pixel tiles (conflict-free A fragment reads)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c28; mkdir -p $O; export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -q -k "conv_halo" ) > $O/gpu_tests.log 2>&1; grep -E "passed|failed|^FAILED" $O/gpu_tests.log | tail -3
CB_MATCH="layer1" timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "halo" | tee $O/conv_bench_pp.txt
timeout 600 bash tools/ab.sh "MN_X=0" "MN_HALO_PP=0" > $O/ab.txt 2>&1; cat $O/ab.txt
rm -f gpurun_out/pmc_conv_l1b.txt; bash tools/pmc_conv.sh "layer1" l1b; grep -A9 "conv_halo_pp" gpurun_out/pmc_conv_l1b.txt | grep -E "halo_pp|BANK_CONFLICT|IDX_ACTIVE" | head -8
