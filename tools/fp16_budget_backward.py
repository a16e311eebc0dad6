"""Is the fp16 build's GRADIENT deviation (profiles/r02/parity_full_size.jsonl: 15.8 % relative L2 over all parameters,
39 % worst tensor at the benchmark shape) the floor of its storage format, or something the kernels add?

CPU only; the oracle is the instrument.  The oracle's own training step (forward, MapNetCriterion, autograd backward) is
replayed with fp16 ROUNDING inserted at exactly the tensors the HIP fp16 build stores in fp16:

  forward   I = input image, W = conv weights (operand copies), Y = raw conv outputs, A = post-activation tensors
  backward  the gradients w.r.t. the same Y and A tensors, multiplied by the loss scale before the rounding and divided
            after it (what the plan does: d(pred) * scale, 1 / scale where gradients enter the fp32 arena); weight
            gradients, BatchNorm statistics, the fc / pose head and the loss stay fp32, as in the build.

and the parameter gradients are compared with the unrounded fp32 oracle's:

    python tools/fp16_budget_backward.py [windows] [H] [W] [loss_scale]

Rows: forward roundings only (gradients fp32) / gradient roundings only / both = what the build does; then the loss scale
swept, which separates rounding (scale-independent) from fp16 underflow of small cotangents (scale-dependent).
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (tooling, not product)


class Store16(torch.autograd.Function):
    """a tensor the fp16 build keeps in fp16: value rounded on the way forward (fwd), its gradient rounded -- under the
    loss scale -- on the way back (bwd)"""

    @staticmethod
    def forward(ctx, t, fwd, bwd, scale):
        ctx.bwd, ctx.scale = bwd, scale
        return t.half().float() if fwd else t

    @staticmethod
    def backward(ctx, g):
        if ctx.bwd:
            g = (g * ctx.scale).half().float() / ctx.scale
        return g, None, None, None


class W16(torch.autograd.Function):
    """fp16 operand copy of an fp32 master weight: rounded forward, gradient passed through (accumulated in fp32)"""

    @staticmethod
    def forward(ctx, t, on):
        return t.half().float() if on else t

    @staticmethod
    def backward(ctx, g):
        return g, None


def bn_train(x, bn):
    return F.batch_norm(x, None, None, bn.weight, bn.bias, True, 0.0, bn.eps)


def forward(net, x, fwd, bwd, scale):
    st = lambda t: Store16.apply(t, fwd, bwd, scale)  # noqa: E731
    fe = net.mapnet.feature_extractor
    n, t = x.shape[:2]
    x = x.reshape(n * t, *x.shape[2:])
    x = x.half().float() if fwd else x
    y = st(F.conv2d(x, W16.apply(fe.conv1.weight, fwd), None, 2, 3))
    a = st(F.max_pool2d(F.relu(bn_train(y, fe.bn1)), 3, 2, 1))  # (the normalised stem activation itself is never stored)
    for li in range(1, 5):
        for blk in getattr(fe, "layer%d" % li):
            y1 = st(F.conv2d(a, W16.apply(blk.conv1.weight, fwd), None, blk.conv1.stride, 1))
            a1 = st(F.relu(bn_train(y1, blk.bn1)))
            y2 = st(F.conv2d(a1, W16.apply(blk.conv2.weight, fwd), None, 1, 1))
            z = bn_train(y2, blk.bn2)
            if blk.downsample is not None:
                yd = st(F.conv2d(a, W16.apply(blk.downsample[0].weight, fwd), None, blk.downsample[0].stride, 0))
                sc = st(bn_train(yd, blk.downsample[1]))
            else:
                sc = a
            a = st(F.relu(z + sc))
    p = a.mean((2, 3))
    feat = F.relu(F.linear(p, fe.fc.weight, fe.fc.bias))
    pn = net.mapnet
    out = torch.cat((F.linear(feat, pn.fc_xyz.weight, pn.fc_xyz.bias), F.linear(feat, pn.fc_wpqr.weight, pn.fc_wpqr.bias)), 1)
    return out.view(n, t, 6)


def grads(net, crit, x, t, fwd, bwd, scale):
    for p in net.parameters():
        p.grad = None
    out = forward(net, x, fwd, bwd, scale)
    loss = crit(out, t)
    loss.backward()
    return loss.item(), out.detach(), {k: v.grad.clone() for k, v in net.mapnet.named_parameters()}


def compare(g, ref):
    num = den = 0.0
    worst, wname = 0.0, ""
    for k, r in ref.items():
        d = (g[k].double() - r.double())
        num += d.pow(2).sum().item()
        den += r.double().pow(2).sum().item()
        if r.norm() > 1e-8:
            e = (d.norm() / r.double().norm()).item()
            if e > worst:
                worst, wname = e, k
    return (num / den) ** 0.5, worst, wname


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 341
    scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1024.0
    torch.manual_seed(7)
    net = oracle.MapNet(oracle.PoseNet(oracle.resnet34(), droprate=0.0, pretrained=False))
    crit = oracle.MapNetCriterion(0.0, -3.0, 0.0, -3.0, True, True)
    x, t = oracle.make_batch("mapnet", n, H, W, seed=7)
    net.train()
    t0 = time.time()
    l0, p0, g0 = grads(net, crit, x, t, False, False, 1.0)
    print("batch %d windows x 3 = %d images %dx%d; fp32 step %.1f s; loss %.4f" % (n, n * 3, H, W, time.time() - t0, l0), flush=True)
    print("%-58s %10s %10s %10s %10s  %s" % ("fp16 rounding at", "loss rel", "pose max", "grad all", "grad worst", "(tensor)"))
    rows = [("forward tensors only (gradients fp32)", True, False, scale),
            ("gradient tensors only, loss scale %g" % scale, False, True, scale),
            ("forward + gradient tensors, loss scale %g (= the build)" % scale, True, True, scale)]
    rows += [("forward + gradient tensors, loss scale %g" % s, True, True, s) for s in (1.0, 32.0, 32768.0) if s != scale]
    for name, fwd, bwd, s in rows:
        l, p, g = grads(net, crit, x, t, fwd, bwd, s)
        a, w, wn = compare(g, g0)
        print("%-58s %10.3e %10.3e %10.3e %10.3e  %s" % (name, abs(l - l0) / max(1.0, abs(l0)), (p - p0).abs().max().item(), a, w, wn),
              flush=True)


if __name__ == "__main__":
    main()
