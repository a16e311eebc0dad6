#!/bin/bash
# Round-2 GPU call 7: 3-slot B ring of the 128-column chunk-resident kernel, element-wise grid sizing, schedule default 2
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c7; mkdir -p $O; export TMPDIR=/tmp
( time timeout 600 python -m pytest tests -m gpu -q -k "conv_forward or data_gradient or chunk_resident or train_step_fp32_parity_small or fp16_close or batchnorm" ) > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log; grep -E "^FAILED|^ERROR" $O/gpu_tests.log | head
for n in 3 2; do echo "--- MN_IGEMM_HALO_NBS=$n"; MN_IGEMM_HALO_NBS=$n CB_MATCH="3x3" timeout 200 python tools/conv_bench.py fp16 192 2>&1 | grep -E "^layer[24] " | sed -e 's/| wgrad.*//'; done | tee $O/conv_bench_nbs.txt
timeout 900 bash tools/ab.sh "MN_X=0" "MN_IGEMM_HALO_NBS=2" "MN_EW_MIN_ITERS=4" "MN_EW_MIN_ITERS=8" "MN_WGRAD_SCHED=1" "MN_WGRAD_SCHED=0" > $O/ab.txt 2>&1; cat $O/ab.txt
