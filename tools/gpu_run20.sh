#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
cat > /tmp/loop.py <<'PY'
import sys, os, time, ctypes as C, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]+"/tests")
import checks
from geomapnet_amd import _binding
from geomapnet_amd._binding import ptr
lib=_binding.hip(); td=torch.float16
g,Ho,Wo=checks.fwd_geom(32,32,32,2048,1024,1,1,0)
x=torch.randn(32,32,32,2048,device="cuda").to(td); w=(torch.randn(1024,2048,device="cuda")*0.02).to(td); y=torch.empty(32,32,32,1024,dtype=td,device="cuda")
one=C.c_float(1.0); zp=checks.zero_page("cuda")
t0=time.time()
while time.time()-t0 < float(sys.argv[1]):
    for _ in range(200): lib.op_igemm(1,C.byref(g),ptr(x),ptr(w),ptr(y),1024,None,None,0,None,None,one,ptr(zp),None)
    torch.cuda.synchronize()
PY
echo "== idle"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | head -6
python /tmp/loop.py 8 &
sleep 5
echo "== under igemm load"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | head -6
wait
./tools/probes/mfma_peak > /dev/null &
sleep 0.2
wait
