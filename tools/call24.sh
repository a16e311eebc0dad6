cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c24; O=$GRAFT_REPO_ROOT/gpurun_out/c24; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/ab.sh "MN_FORK_EVENT_SCOPE=1" "MN_FORK_EVENT_SCOPE=0" "MN_FORK_EVENT_SCOPE=2" 2>&1 | tee $O/ab.txt
DT=fp16 bash tools/ab.sh "MN_FORK_EVENT_SCOPE=1" "MN_FORK_EVENT_SCOPE=0" 2>&1 | tee -a $O/ab.txt
LEAN="--steps 8 --warmup 3 --repeats 1 --no-cpu-baseline --no-events --no-fast-mode --no-eval-metric --no-feed"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_ovl -o r -- python $R/bench.py $LEAN > $O/rocprof_ovl.log 2>&1
python3 $R/tools/gap_analysis.py /tmp/p_ovl/r_kernel_trace.csv 11 > $O/gaps_overlapped_device_scope.txt 2>&1
head -8 $O/gaps_overlapped_device_scope.txt
cd $R && timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "step or train or mapnet" 2>&1 | grep -E "passed|failed|error" | tee $O/parity.txt
