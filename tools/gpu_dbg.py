import sys, os, torch, ctypes
mode = sys.argv[1]
root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
torch.zeros(1).cuda()
from geomapnet_amd import _binding
lib = _binding.hip()
open(os.path.join(root, "gpurun_out", "maps_%s.txt" % mode), "w").write(open("/proc/self/maps").read())
import checks
if mode == "ops":
    checks.check_conv_fwd(lib, 'cuda', 1, 2, 9, 11, 64, 64, 3, 1, 1); print("igemm ok", flush=True)
    checks.check_adam(lib, 'cuda', n=1000); print("adam ok", flush=True)
else:
    import geomapnet_amd as G, oracle
    G.set_compute_dtype('fp16')
    net = G.MapNet(G.PoseNet(G.resnet34(), droprate=0.0, pretrained=False)).cuda()
    x, t = oracle.make_batch('mapnet', 2, 64, 85); x = x.cuda()
    net.eval(); y = net(x); torch.cuda.synchronize(); print("fwd ok", flush=True)
