"""How far apart do two runs of the ORACLE drift over consecutive training steps?  (CPU only.)

The parity tests hold the first step of the HIP path to the north-star bar (loss 1e-4 relative, pose 1e-3) and later steps
to a loose one (tests/checks.py: check_train_step).  This tool shows why: the reference's own step -- random-init ResNet-34,
BatchNorm in training mode, Adam's sign-like first updates -- amplifies a one-ulp perturbation of the weights, or just a
different summation order (thread count), by ~30x per step.  usage: python tools/oracle_sensitivity.py [windows H W steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402


def run(threads, N, H, W, steps, perturb=0.0):
    torch.set_num_threads(threads)
    torch.manual_seed(7)
    net = oracle.MapNet(oracle.PoseNet(oracle.resnet34(), droprate=0.0, pretrained=False))
    if perturb:
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(1.0 + perturb)
    x, t = oracle.make_batch("mapnet", N, H, W, seed=7)
    crit = oracle.MapNetCriterion(0.0, -3.0, 0.0, -3.0, True, True)
    groups = [{"params": net.parameters()}, {"params": [crit.sax, crit.saq]}, {"params": [crit.srx, crit.srq]}]
    opt = oracle.Optimizer(groups, "adam", base_lr=1e-4, weight_decay=5e-4)
    net.train()
    out = []
    for _ in range(steps):
        loss, poses = oracle.step_feedfwd(x, net, False, t, crit, opt, True, 0.0)
        out.append((loss, poses.detach().clone()))
    return out


def main():
    N, H, W, steps = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (2, 64, 85, 3)
    a, b, c = run(1, N, H, W, steps), run(8, N, H, W, steps), run(8, N, H, W, steps, 1e-7)
    print("oracle vs oracle, MapNet %d windows x 3 x %dx%d, Adam lr 1e-4" % (N, H, W))
    for s in range(steps):
        print("step %d  1 vs 8 threads: loss rel %.2e pose max %.2e | weights * (1 + 1e-7): loss rel %.2e pose max %.2e"
              % (s + 1, abs(a[s][0] - b[s][0]) / max(1, abs(a[s][0])), (a[s][1] - b[s][1]).abs().max().item(),
                 abs(b[s][0] - c[s][0]) / max(1, abs(b[s][0])), (b[s][1] - c[s][1]).abs().max().item()))


if __name__ == "__main__":
    main()
