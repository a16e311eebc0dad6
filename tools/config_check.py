"""usage: python tools/config_check.py [dtype ...]   (default: fp16 fp16x2m)
Runs the other BASELINE.json configurations at their FULL sizes for a few steps (synthetic data): PoseNet
batch 64 (configs[1]) and MapNet++ 64 windows x 2T = 384 images with MapNetOnlineCriterion, max_grad_norm 5 and the NaN
filter (configs[4], per-GPU share).  Prints images/s; they are parity-test shapes, not bench lines."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import geomapnet_amd as G  # noqa: E402
import oracle  # noqa: E402


def run(mode, n, steps=6, dtype="fp16"):
    G.set_compute_dtype(dtype)
    torch.manual_seed(7)
    pn = G.PoseNet(G.resnet34(), droprate=0.0, pretrained=False, filter_nans=(mode == "mapnet++"))
    if mode == "posenet":
        net = pn.cuda()
        crit = G.PoseNetCriterion(sax=0.0, saq=-3.0, learn_beta=True).cuda()
        groups = [{"params": net.parameters()}, {"params": [crit.sax, crit.saq]}]
        lr, wd, clip = 1e-4, 5e-4, 0.0
    else:
        net = G.MapNet(pn).cuda()
        crit = G.MapNetOnlineCriterion(sax=0.0, saq=-3.0, srx=0.0, srq=-3.0, learn_beta=True, learn_gamma=True).cuda()
        groups = [{"params": net.parameters()}, {"params": [crit.sax, crit.saq]}, {"params": [crit.srx, crit.srq]}]
        lr, wd, clip = 1e-5, 0.0, 5.0
    opt = G.Optimizer(groups, "adam", base_lr=lr, weight_decay=wd)
    net.train()
    x, t = oracle.make_batch(mode, n, 256, 341, seed=7)
    x, t = x.cuda(), t.cuda()
    images = x.shape[0] * (x.shape[1] if x.dim() == 5 else 1)
    losses = []
    for i in range(steps + 2):
        if i == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        l, out = G.step_feedfwd(x, net, True, t, crit, opt, True, clip)
        losses.append(l)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert all(l == l and abs(l) < 1e6 for l in losses), losses
    print("%-8s %-9s %3d images/step: %.2f ms/step = %.0f images/s; loss %.3f -> %.3f; output %s"
          % (dtype, mode, images, dt * 1e3, images / dt, losses[0], losses[-1], tuple(out.shape)), flush=True)


if __name__ == "__main__":
    for dt in (sys.argv[1:] or ["fp16", "fp16x2m"]):
        run("posenet", 64, dtype=dt)
        run("mapnet++", 64, dtype=dt)
