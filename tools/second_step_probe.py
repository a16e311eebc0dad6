"""GPU: how far is the HIP path's 2nd / 3rd training step from the oracle's -- and how far is the fp32 oracle from ITSELF run
in fp64?  (diagnostic behind tests/test_gpu_parity.py::test_second_step_fp32_with_a_smooth_update and DESIGN.md section 6)
usage: python tools/second_step_probe.py [fp32|fp16] [lr] [adam_eps] [steps]"""
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

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
eps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = "cuda" if torch.cuda.is_available() else "cpu"
if dev == "cuda":
    lib = _binding.hip()
else:
    import emu_lib
    lib = emu_lib.load()
G.set_compute_dtype(dtype)
x, t = oracle.make_batch("mapnet", 2, 64, 85, seed=7)


def oracle_run(double):
    net, _ = checks.build_pair(lib, "cpu")
    crit = oracle.MapNetCriterion(0.0, -3.0, 0.0, -3.0, True, True)
    xx, tt = x, t
    if double:
        net, crit, xx, tt = net.double(), crit.double(), x.double(), t.double()
    groups = [{"params": net.parameters()}, {"params": [crit.sax, crit.saq]}, {"params": [crit.srx, crit.srq]}]
    opt = oracle.Optimizer(groups, "adam", base_lr=lr, weight_decay=5e-4, eps=eps)
    net.train()
    out = []
    for _ in range(steps):
        loss, poses = oracle.step_feedfwd(xx, net, False, tt, crit, opt, True, 0.0)
        out.append((float(loss), poses.detach().double().clone()))
    return out


def hip_run():
    _, net = checks.build_pair(lib, dev)
    crit = G.MapNetCriterion(sax=0.0, saq=-3.0, srx=0.0, srq=-3.0, learn_beta=True, learn_gamma=True, _binding=lib)
    groups = [{"params": net.parameters()}, {"params": [crit.sax, crit.saq]}, {"params": [crit.srx, crit.srq]}]
    opt = G.Optimizer(groups, "adam", base_lr=lr, weight_decay=5e-4, eps=eps)
    net.train()
    out = []
    for _ in range(steps):
        loss, poses = G.step_feedfwd(x.to(dev), net, dev != "cpu", t.to(dev), crit, opt, True, 0.0)
        out.append((float(loss), poses.detach().cpu().double().clone()))
    return out


o32, o64, hip = oracle_run(False), oracle_run(True), hip_run()
knobs = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("MN_"))
print("[%s] %s build, Adam lr %g eps %g; reference = the oracle run in fp64" % (knobs, dtype, lr, eps))
for s in range(steps):
    def d(a, b):
        return abs(a[s][0] - b[s][0]) / max(1.0, abs(b[s][0])), (a[s][1] - b[s][1]).abs().max().item()
    print("step %d  loss(fp64) %.6f | fp32 oracle vs fp64: loss rel %.2e pose max %.2e | HIP vs fp64: loss rel %.2e pose max %.2e | "
          "HIP vs fp32 oracle: loss rel %.2e pose max %.2e" % ((s + 1, o64[s][0]) + d(o32, o64) + d(hip, o64) + d(hip, o32)))
