cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c14; O=$GRAFT_REPO_ROOT/gpurun_out/c14
T0=$(date +%s)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
echo "rc=$? wall_s=$(( $(date +%s) - T0 ))" | tee $O/wall.txt
tail -1 $O/bench_driver_command.json | cut -c1-600
