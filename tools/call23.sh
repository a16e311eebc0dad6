cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c23; O=$GRAFT_REPO_ROOT/gpurun_out/c23; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
LEAN="--steps 8 --warmup 3 --repeats 1 --no-cpu-baseline --no-events --no-fast-mode --no-eval-metric --no-feed"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_ovl -o r -- python $R/bench.py $LEAN > $O/rocprof_ovl.log 2>&1
python3 $R/tools/gap_analysis.py /tmp/p_ovl/r_kernel_trace.csv 11 > $O/gaps_overlapped.txt 2>&1
cd /tmp && MN_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_ser -o r -- python $R/bench.py $LEAN > $O/rocprof_ser.log 2>&1
python3 $R/tools/gap_analysis.py /tmp/p_ser/r_kernel_trace.csv 11 > $O/gaps_serial.txt 2>&1
head -c 3000 $O/gaps_overlapped.txt; head -30 $O/gaps_serial.txt
python3 - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/p_ovl/r_kernel_trace.csv')))
print(rows[0].keys())
PY
