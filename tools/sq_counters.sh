#!/bin/bash
# SQ / GRBM counters of every kernel of a training step on the final tree: per kernel the matrix-pipe busy
# fraction (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE)), LDS stalls and bank conflicts.
#   gpurun -- 'TAG=c1 bash tools/sq_counters.sh'  ->  gpurun_out/sq_$TAG/sq_counters_<dtype>.txt (+ .json)
# Counter passes are their own runs with --kernel-trace only (no --stats / sys-trace), serial kernels (MN_WGRAD_STREAM=0).
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sq_${TAG:-cur}; mkdir -p $O; export TMPDIR=/tmp
SETS=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
      "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE")
for dt in ${DTYPES:-fp16 fp16x2m}; do
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    rm -rf /tmp/sq_$dt_$i
    (cd /tmp && MN_WGRAD_STREAM=0 timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/sq_${dt}_$i -o r -- python $R/bench.py --dtype $dt --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-events --no-parity-mode --no-eval-metric --no-feed > $O/log_${dt}_$i.txt 2>&1)
  done
  python3 $R/tools/sq_counters_table.py $dt /tmp/sq_${dt}_1/r_counter_collection.csv /tmp/sq_${dt}_2/r_counter_collection.csv /tmp/sq_${dt}_1/r_kernel_trace.csv $O
done
