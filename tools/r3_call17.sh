#!/bin/bash
# round 3, GPU call 17: default bench record, kernel stats (overlapped / serial / fp32x3), PMC traffic; fp16 train step N=2 errors
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r3c17 PMC=1 X3=1 bash tools/gpu_prof.sh
cd "$GRAFT_REPO_ROOT"
python3 - <<'PY' 2>&1 | tail -5 | tee gpurun_out/prof_r3c17/fp16_step_n2.txt
import sys; sys.path[:0]=['.','tests']
import checks
from geomapnet_amd import _binding
lib=_binding.hip()
for N,H,W in ((2,256,341),(2,64,85),(4,256,341)):
    rep=checks.check_train_step(lib,"cuda","fp16",mode="mapnet",N=N,H=H,W=W,steps=1,loss_rtol=5e-2,pose_atol=1e-1,grad_l2_rtol=None)
    l,lo,pe=rep[0]; print("fp16 step N=%d %dx%d: loss %.6f oracle %.6f rel %.3e pose max abs %.3e"%(N,H,W,l,lo,abs(l-lo)/max(1,abs(lo)),pe))
PY
