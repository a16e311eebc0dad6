"""One-GPU rehearsal of the 8-GPU data-parallel step: step time against the number of co-resident
"collective" workgroups.  The staged (DP) step is issued exactly as geomapnet_amd/dp.py issues it under torch.distributed, but each
gradient bucket's all-reduce is replaced by the library's occupancy stand-in (mn_op_occupy, csrc/rehearsal.h): c workgroups x 256
threads resident on a communication stream for the time a ring all-reduce of that bucket over 8 GPUs would take (200 GB/s bus
bandwidth + 40 us), streaming the bucket's reduce traffic through HBM meanwhile.

    python tools/rccl_rehearsal.py [fp16x2m fp16]   ->  table: ms/step by (c, MN_DP_DEFER schedule), two interleaved passes

c = 0 is the staged step with no collective at all (the one-GPU cost of issuing the step in stages)."""
import os
import sys
import time

os.environ["MN_FORCE_STAGED"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import geomapnet_amd as G  # noqa: E402
from geomapnet_amd import dp  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
STEPS = int(os.environ.get("STEPS", "30"))
CS = [int(v) for v in os.environ.get("CS", "0,4,8,16,32,64").split(",")]
THREADS = os.environ.get("THREADS", "256")
BUSBW = os.environ.get("BUSBW", "200")
LDS_KB = os.environ.get("LDS_KB", "0")


def build(dtype):
    G.set_compute_dtype(dtype)
    torch.manual_seed(7)
    net = G.MapNet(G.PoseNet(G.resnet34(), droprate=0.0, pretrained=False))
    crit = G.MapNetCriterion(sax=0.0, saq=-3.0, srx=0.0, srq=-3.0, learn_beta=True, learn_gamma=True)
    net.cuda()
    crit.cuda()
    opt = G.Optimizer([{"params": net.parameters()}, {"params": [crit.sax, crit.saq]}, {"params": [crit.srx, crit.srq]}],
                      "adam", base_lr=1e-4, weight_decay=5e-4)
    net.train()
    gen = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(64, 3, 3, 256, 341, device=dev, generator=gen)
    t = torch.randn(64, 3, 6, device=dev, generator=gen) * 0.3
    return net, crit, opt, x, t


def region(net, crit, opt, x, t, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        G.step_feedfwd(x, net, True, t, crit, opt, True)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


for dtype in (sys.argv[1:] or ["fp16x2m", "fp16"]):
    net, crit, opt, x, t = build(dtype)
    for _ in range(8):
        G.step_feedfwd(x, net, True, t, crit, opt, True)
    print("== %s: staged step, stand-in collectives of c workgroups x %s threads x %s KB of LDS (ring model: %d GPUs, %s GB/s busbw + 40 us: "
          "%s us for the 57 / 27 / 4.5 / 0.9 MB buckets)"
          % (dtype, THREADS, LDS_KB, dp.RING_WORLD, BUSBW,
             " / ".join("%.0f" % dp.ring_allreduce_us(b, dp.RING_WORLD, float(BUSBW)) for b in (57e6, 27e6, 4.5e6, 0.9e6))), flush=True)
    table = {}
    for rep in range(2):
        for defer in (0, 1, 2):
            for c in CS:
                if c == 0 and defer != 0:
                    continue
                os.environ["MN_DP_DEFER"] = str(defer)
                if c:
                    os.environ["MN_DP_STANDIN"] = "%d,%s,%s,40,%s" % (c, THREADS, BUSBW, LDS_KB)
                else:
                    os.environ.pop("MN_DP_STANDIN", None)
                region(net, crit, opt, x, t, 3)
                table.setdefault((defer, c), []).append(region(net, crit, opt, x, t, STEPS))
    os.environ.pop("MN_DP_STANDIN", None)
    print("%-44s" % "schedule \\ c" + "".join("%16d" % c for c in CS))
    names = {0: "0: each bucket after its stage (default)", 1: "1: bucket 3 at once, 2..0 after stage 0", 2: "2: all after the last stage"}
    for defer in (0, 1, 2):
        row = "%-44s" % names[defer]
        for c in CS:
            v = table.get((defer, c))
            row += "%16s" % ("%.2f / %.2f" % tuple(v) if v else "-")
        print(row, flush=True)
    del net, crit, opt, x, t
    torch.cuda.empty_cache()
