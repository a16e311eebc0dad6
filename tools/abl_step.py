"""Whole-step TIMING experiments with the ablation build (tools/ablation/libmapnet_hip_abl.so, results wrong by construction):
the bench workload (64 windows x T=3, 256x341, MapNet criterion, Adam) with learning rate 0, so that an experiment which
leaves stale state behind (e.g. MN_ABL_SKIP_FINALIZE: BatchNorm coefficients of the warm-up steps) keeps every value finite.
usage: MN_LIB=tools/ablation/libmapnet_hip_abl.so [knobs] python tools/abl_step.py [dtype] [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import geomapnet_amd as G  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
G.set_compute_dtype(dtype)
torch.manual_seed(3)
net = G.MapNet(G.PoseNet(G.resnet34(), droprate=0.0, pretrained=False))
crit = G.MapNetCriterion(sax=0.0, saq=-3.0, srx=0.0, srq=-3.0, learn_beta=True, learn_gamma=True)
net.cuda()
crit.cuda()
opt = G.Optimizer([{"params": net.parameters()}, {"params": [crit.sax, crit.saq]}, {"params": [crit.srx, crit.srq]}], "adam",
                  base_lr=0.0, weight_decay=0.0)
net.train()
x = torch.randn(64, 3, 3, 256, 341).cuda()
t = (torch.randn(64, 3, 6) * 0.5).cuda()
for _ in range(12):
    G.step_feedfwd(x, net, True, t, crit, opt, True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    G.step_feedfwd(x, net, True, t, crit, opt, True)
torch.cuda.synchronize()
print("%s %s: %.3f ms/step" % (dtype, " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("MN_") and k != "MN_LIB"),
                               (time.perf_counter() - t0) / steps * 1e3))
