cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c7; O=$GRAFT_REPO_ROOT/gpurun_out/c7
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "weights_in_registers" > $O/test_halo_h2.txt 2>&1; tail -3 $O/test_halo_h2.txt
timeout 300 python tools/halo_h2_bench.py > $O/halo_h2_bench.txt 2>&1; tail -2 $O/halo_h2_bench.txt
for w in 256 512; do MN_HALO_H2_WGS=$w timeout 300 python tools/halo_h2_bench.py 2>&1 | tail -1 | sed "s/^/[wgs $w] /" >> $O/halo_h2_bench.txt; done
DT=fp16x2m STEPS=40 bash tools/ab.sh "MN_HALO_H2=0" "MN_HALO_H2=1" > $O/ab_halo_h2_fp16x2m.txt 2>&1; cat $O/ab_halo_h2_fp16x2m.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fp16x2m or fp16x2 or configs2" > $O/tests_mixed.txt 2>&1; tail -3 $O/tests_mixed.txt
