"""First-step parameter gradients of the HIP path (fp32 or fp16 build) and of the fp32 oracle, each against the oracle run in
fp64: overall and per-tensor relative L2 error.  Runs on the GPU (or, without one, through the emulator).
usage: python tools/grad_accuracy.py [fp32|fp16] [windows H W]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import checks  # noqa: E402
import oracle  # noqa: E402
import geomapnet_amd as G  # noqa: E402
from geomapnet_amd import _binding  # noqa: E402
from geomapnet_amd.posenet import _view  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
N, H, W = [int(v) for v in sys.argv[2:5]] if len(sys.argv) >= 5 else (2, 64, 85)
dev = "cuda" if torch.cuda.is_available() else "cpu"
if dev == "cuda":
    lib = _binding.hip()
else:
    import emu_lib
    lib = emu_lib.load()
G.set_compute_dtype(dtype)
x, t = oracle.make_batch("mapnet", N, H, W, seed=7)


def oracle_grads(double):
    net, _ = checks.build_pair(lib, "cpu")
    crit = oracle.MapNetCriterion(0.0, -3.0, 0.0, -3.0, True, True)
    xx, tt = x, t
    if double:
        net, crit, xx, tt = net.double(), crit.double(), x.double(), t.double()
    net.train()
    loss = crit(net(xx), tt)
    loss.backward()
    return float(loss), {k: v.grad.double() for k, v in net.named_parameters()}


l64, g64 = oracle_grads(True)
l32, g32 = oracle_grads(False)
_, net = checks.build_pair(lib, dev)
crit = G.MapNetCriterion(sax=0.0, saq=-3.0, srx=0.0, srq=-3.0, learn_beta=True, learn_gamma=True, _binding=lib)
opt = G.Optimizer([{"params": net.parameters()}, {"params": [crit.sax, crit.saq]}, {"params": [crit.srx, crit.srq]}], "adam",
                  base_lr=1e-12, weight_decay=0.0)
net.train()
loss, _ = G.step_feedfwd(x.to(dev), net, dev != "cpu", t.to(dev), crit, opt, True, 0.0)
eng = net.mapnet._engine
scale = eng.loss_scale_state()[0] if dtype == "fp16" else 1.0
rows, tot = [], {"hip": [0.0, 0.0], "o32": [0.0, 0.0]}
for e in eng.entries:
    if e.is_buffer:
        continue
    name = "mapnet." + e.name.decode()
    r = g64[name]
    if float(r.norm()) < 1e-12:
        continue
    gh = _view(eng.grads(), e).cpu().double()
    eh, eo = float((gh - r).norm() / r.norm()), float((g32[name] - r).norm() / r.norm())
    rows.append((eh, eo, name, float(r.norm())))
    for key, g in (("hip", gh), ("o32", g32[name])):
        tot[key][0] += float((g - r).pow(2).sum())
        tot[key][1] += float(r.pow(2).sum())
print("%s build on %s, MapNet %d x 3 x %dx%d: loss fp64 %.6f, fp32 oracle %.6f, HIP %.6f" % (dtype, dev, N, H, W, l64, l32, float(loss)))
print("all parameters, relative L2 against the fp64 oracle: HIP %.3e   fp32 oracle %.3e"
      % ((tot["hip"][0] / tot["hip"][1]) ** 0.5, (tot["o32"][0] / tot["o32"][1]) ** 0.5))
rows.sort(reverse=True)
print("worst tensors of the HIP path:   HIP        fp32 oracle")
for eh, eo, name, nr in rows[:15]:
    print("  %-52s %.3e  %.3e  |g| %.2e" % (name, eh, eo, nr))
print("median tensor: HIP %.3e" % rows[len(rows) // 2][0])
if os.environ.get("GA_ALL"):  # every tensor, in the engine's (network) order
    order = {("mapnet." + e.name.decode()): i for i, e in enumerate(eng.entries)}
    for eh, eo, name, nr in sorted(rows, key=lambda r: order[r[2]]):
        print("  %-52s %.3e  %.3e  |g| %.2e" % (name, eh, eo, nr))
