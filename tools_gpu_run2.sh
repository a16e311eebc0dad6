#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
for mode in ops fwd; do
  echo "== rocprofv3 kernel-trace+stats $mode"
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_$mode -o t -- python $GRAFT_REPO_ROOT/tools_gpu_dbg.py $mode > $GRAFT_REPO_ROOT/gpurun_out/dbg_$mode.log 2>&1
  echo "rc=$?"; grep -v "^W2026\|^E2026" $GRAFT_REPO_ROOT/gpurun_out/dbg_$mode.log | head -30 | cut -c1-200
done
echo "== kernel-trace only (no stats), fwd"
timeout 300 rocprofv3 --kernel-trace -d /tmp/p_kt -o t -- python $GRAFT_REPO_ROOT/tools_gpu_dbg.py fwd > $GRAFT_REPO_ROOT/gpurun_out/dbg_kt.log 2>&1; echo "rc=$?"
echo "== hip-trace fwd"
timeout 300 rocprofv3 --hip-runtime-trace --stats -d /tmp/p_hip -o t -- python $GRAFT_REPO_ROOT/tools_gpu_dbg.py fwd > $GRAFT_REPO_ROOT/gpurun_out/dbg_hip.log 2>&1; echo "rc=$?"
ls -R /tmp/p_* 2>/dev/null | head -30
