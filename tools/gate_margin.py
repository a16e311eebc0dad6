"""CPU: how close to zero do the ReLU inputs of the oracle come, and does fp32 evaluation flip any of their signs?
One flipped gate changes every gradient below it discontinuously: with 36 spatial positions x 512 channels in layer4 a single
flip is ~1 % of a tensor's gradient norm -- the size of the fp32 HIP build's first-step gradient distance on the GPU
(tools/grad_accuracy.py), while the forward pass agrees to 1e-6.  usage: python tools/gate_margin.py [windows H W]"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

N, H, W = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else (2, 64, 85)
pre = {}
orig_relu = F.relu


def run(double):
    torch.manual_seed(7)
    net = oracle.MapNet(oracle.PoseNet(oracle.resnet34(), droprate=0.0, pretrained=False))
    x, _ = oracle.make_batch("mapnet", N, H, W, seed=7)
    if double:
        net, x = net.double(), x.double()
    net.train()
    rec = []

    def relu(v, inplace=False):  # every ReLU of the oracle goes through F.relu (the stem's nn.ReLU too)
        rec.append(v.detach().double().clone())
        return orig_relu(v)

    F.relu = relu
    try:
        with torch.no_grad():
            net(x)
    finally:
        F.relu = orig_relu
    return rec


a64, a32 = run(True), run(False)
print("ReLU inputs of the oracle, MapNet %d x 3 x %dx%d: elements within 1e-5 / 1e-6 of zero (fp64), sign flips fp32 vs fp64" % (N, H, W))
tot = 0
for i, (v64, v32) in enumerate(zip(a64, a32)):
    flips = int(((v64 > 0) != (v32 > 0)).sum())
    tot += flips
    print("  relu %2d  %9d elements   |v|<1e-5: %4d   |v|<1e-6: %3d   flips: %d   rms %.2f"
          % (i, v64.numel(), int((v64.abs() < 1e-5).sum()), int((v64.abs() < 1e-6).sum()), flips, float(v64.pow(2).mean().sqrt())))
print("total sign flips between the fp32 and the fp64 oracle:", tot)
