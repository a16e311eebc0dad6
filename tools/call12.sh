cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c12; O=$GRAFT_REPO_ROOT/gpurun_out/c12
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "weights_in_registers" > $O/test_halo_h2.txt 2>&1; tail -2 $O/test_halo_h2.txt
timeout 300 python tools/halo_h2_bench.py 2>&1 | tail -1 > $O/halo_h2_bench.txt
MN_HALO_H2_FORM=4 timeout 300 python tools/halo_h2_bench.py 2>&1 | tail -1 | sed "s/^/[4-wave form] /" >> $O/halo_h2_bench.txt
for a in 0 300 1 2 4 5 7 16; do MN_LIB=$GRAFT_REPO_ROOT/tools/ablation/libmapnet_hip_abl.so MN_HALO_H2_ABLATE=$a timeout 300 python tools/halo_h2_bench.py 2>&1 | tail -1 | sed "s/^/[8-wave variant $a] /" >> $O/halo_h2_bench.txt; done
cat $O/halo_h2_bench.txt | cut -c1-185
DT=fp16x2m STEPS=40 bash tools/ab.sh "MN_HALO_H2=0" "MN_HALO_H2_FORM=4" "MN_HALO_H2_FORM=8" > $O/ab_halo_h2_fp16x2m.txt 2>&1; cat $O/ab_halo_h2_fp16x2m.txt
