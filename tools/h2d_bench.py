"""PCIe-inclusive step rate: each step copies its batch from pinned host memory, as the reference's loop does
(common/train.py:341-347), for fp32 NCHW input and for uint8 NHWC input with device-side normalisation.
Not the headline metric (bench.py times steps with inputs resident in HBM).  usage: python tools/h2d_bench.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import geomapnet_amd as G  # noqa: E402

N, T, H, W = 64, 3, 256, 341
dev = torch.device("cuda", 0)
G.set_compute_dtype("fp16")
torch.manual_seed(7)
net = G.MapNet(G.PoseNet(G.resnet34(), droprate=0.0, pretrained=False)).cuda()
crit = G.MapNetCriterion(sax=0.0, saq=-3.0, srx=0.0, srq=-3.0, learn_beta=True, learn_gamma=True).cuda()
opt = G.Optimizer([{"params": net.parameters()}, {"params": [crit.sax, crit.saq]}, {"params": [crit.srx, crit.srq]}], "adam",
                  base_lr=1e-4, weight_decay=5e-4)
net.train()
targets = torch.randn(N, T, 6).pin_memory()
mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
for name in ("fp32 NCHW", "uint8 NHWC"):
    if name.startswith("uint8"):
        net.set_input_u8(mean, std)
        host = [torch.randint(0, 256, (N, T, H, W, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
    else:
        host = [torch.randn(N, T, 3, H, W).pin_memory() for _ in range(2)]
    def step(i):
        x = host[i % 2].to(dev, non_blocking=True)
        t = targets.to(dev, non_blocking=True)
        return G.step_feedfwd(x, net, True, t, crit, opt, True)[0]
    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for i in range(K):
        step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("%-11s host->device %6.1f MB/step: %7.2f ms/step  %8.1f images/s (copy + step, serial as in the reference loop)"
          % (name, host[0].numel() * host[0].element_size() / 1e6, dt * 1e3, N * T / dt), flush=True)
