#!/bin/bash
# round 3, GPU call 12: igemm_halo K-loop ablation (DMA / fragment reads / MFMA); serial per-dispatch kernel trace by grid size
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c12; mkdir -p $O
ABL=$PWD/tools/ablation/libmapnet_hip_abl.so
for a in 0 4 8 16 12 20 24 28; do
  echo "== MN_HALO_ABLATE=$a (4 no DMA in the K loop, 8 fragment reads only in the first K-step, 16 no MFMAs)" >> $O/igemm_halo_kloop_ablation.txt
  MN_HALO_ABLATE=$a MN_LIB=$ABL timeout 200 python tools/conv_bench.py fp16 2>&1 | grep -E "^layer(2|3|4) 3x3 (128|256|512)" | grep -v "through the workspace" | cut -c1-120 >> $O/igemm_halo_kloop_ablation.txt
done
cat $O/igemm_halo_kloop_ablation.txt
export TMPDIR=/tmp
cd /tmp && MN_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_serial -o r -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-events --no-parity-mode > $O/rocprof.log 2>&1
python3 - <<PY
import csv, collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open('/tmp/prof_serial/r_kernel_trace.csv')):
    agg[(r['Kernel_Name'][:80], r['Grid_Size_X'], r['Workgroup_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
with open('$O/serial_by_grid.csv','w') as f:
    w=csv.writer(f); w.writerow(['kernel','grid_x','wg_x','dispatches','mean_us','min_us','total_us'])
    for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
        w.writerow([k[0],k[1],k[2],len(v),round(sum(v)/len(v),1),round(min(v),1),round(sum(v),1)])
PY
head -70 $O/serial_by_grid.csv
