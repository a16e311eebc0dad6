cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c25; O=$GRAFT_REPO_ROOT/gpurun_out/c25; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
LEAN="--no-cpu-baseline --no-events --no-fast-mode --no-eval-metric --no-feed"
# first: does each mode run at all (own timeout: a wait that is never satisfied must not hang the box)
for m in 1 2; do
  MN_FORK_MODE=$m timeout 240 python bench.py --steps 10 --warmup 3 --repeats 1 $LEAN > $O/mode$m.json 2> $O/mode$m.err; echo "mode $m rc=$?" | tee -a $O/modes.txt
  tail -1 $O/mode$m.json | cut -c1-200
done
if grep -q "mode 1 rc=0" $O/modes.txt; then A1="MN_FORK_MODE=1"; fi
if grep -q "mode 2 rc=0" $O/modes.txt; then A2="MN_FORK_MODE=2"; fi
bash tools/ab.sh "MN_FORK_MODE=0" $A1 $A2 2>&1 | tee $O/ab.txt
