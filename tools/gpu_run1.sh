#!/bin/bash
# GPU session: sanity -> parity tests -> bench -> rocprof kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== sanity" 
timeout 300 python - > gpurun_out/sanity.log 2>&1 <<'PY'
import torch, time, sys, ctypes
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
print(torch.cuda.get_device_name(0), torch.version.hip)
from geomapnet_amd import _binding
lib=_binding.hip(); print('backend', lib.backend_name)
hip = ctypes.CDLL("libamdhip64.so")
import geomapnet_amd as G, oracle
G.set_compute_dtype('fp32')
net = G.MapNet(G.PoseNet(G.resnet34(), droprate=0.0, pretrained=False)).cuda()
x,t = oracle.make_batch('mapnet', 2, 64, 85)
x=x.cuda(); t=t.cuda()
print('stale hip error before first call:', hip.hipGetLastError())
net.eval(); y = net(x); torch.cuda.synchronize(); print('eval fwd ok', y.abs().max().item())
net.train(); y = net(x); torch.cuda.synchronize(); print('train fwd ok', y.abs().max().item())
import checks
print(checks.check_train_step(lib,'cuda','fp32',N=2,H=64,W=85))
print(checks.check_train_step(lib,'cuda','fp16',N=2,H=64,W=85,loss_rtol=2e-2,pose_atol=5e-2,grad_l2_rtol=None))
PY
tail -8 gpurun_out/sanity.log
echo "== gpu tests"
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/gpu_tests.log 2>&1
tail -30 gpurun_out/gpu_tests.log
echo "== bench"
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1
tail -3 gpurun_out/bench.log
echo "== rocprof"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -o r1 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof1 | head -20
find gpurun_out/prof1 -name "*kernel_stats*" | head -1 | xargs -I{} head -30 {}
find gpurun_out/prof1 -size +20M -delete
