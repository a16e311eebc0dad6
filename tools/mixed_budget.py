"""What does a SINGLE-fp16 backward cost the parity mode's gradients?  (round 5: the `fp16x2m` mode)

CPU only; the oracle is the instrument.  The oracle's training step is replayed with an EXACT fp32 forward pass (what the
fp16x2 forward delivers to 2^-22) and a backward pass whose CONTRACTIONS take fp16 operands, as one-MFMA kernels would:

  data gradient     d(input) = conv_transpose(fp16(dY * scale), fp16(W)) / scale
  weight gradient   dW       = conv_weight(fp16(X), fp16(dY * scale)) / scale

(products exact, fp32 accumulation; X = the stored activation's hi half, dY = the BatchNorm backward's output).  Optional third
argument of a row: the gradients w.r.t. activations (the data gradients' outputs) are ALSO stored in fp16.

    python tools/mixed_budget.py [windows] [H] [W] [loss_scale]
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (tooling, not product)
from tools.fp16_budget_backward import compare, bn_train  # noqa: E402

MODE = {"ops": False, "gact": False, "scale": 1024.0}


def r16(t):
    return t.half().float()


class Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, pad):
        ctx.save_for_backward(x, w)
        ctx.stride, ctx.pad = stride, pad
        return F.conv2d(x, w, None, stride, pad)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        s = MODE["scale"]
        if MODE["ops"]:
            g16 = r16(g * s)
            gx = torch.nn.grad.conv2d_input(x.shape, r16(w), g16, ctx.stride, ctx.pad) / s if ctx.needs_input_grad[0] else None
            gw = torch.nn.grad.conv2d_weight(r16(x), w.shape, g16, ctx.stride, ctx.pad) / s
        else:
            gx = torch.nn.grad.conv2d_input(x.shape, w, g, ctx.stride, ctx.pad) if ctx.needs_input_grad[0] else None
            gw = torch.nn.grad.conv2d_weight(x, w.shape, g, ctx.stride, ctx.pad)
        if gx is not None and MODE["gact"]:
            gx = r16(gx * s) / s
        return gx, gw, None, None


def forward(net, x):
    fe = net.mapnet.feature_extractor
    n, t = x.shape[:2]
    x = x.reshape(n * t, *x.shape[2:])
    y = Conv.apply(x, fe.conv1.weight, 2, 3)
    a = F.max_pool2d(F.relu(bn_train(y, fe.bn1)), 3, 2, 1)
    for li in range(1, 5):
        for blk in getattr(fe, "layer%d" % li):
            st = blk.conv1.stride[0]
            y1 = Conv.apply(a, blk.conv1.weight, st, 1)
            a1 = F.relu(bn_train(y1, blk.bn1))
            y2 = Conv.apply(a1, blk.conv2.weight, 1, 1)
            z = bn_train(y2, blk.bn2)
            if blk.downsample is not None:
                sc = bn_train(Conv.apply(a, blk.downsample[0].weight, st, 0), blk.downsample[1])
            else:
                sc = a
            a = F.relu(z + sc)
    p = a.mean((2, 3))
    feat = F.relu(F.linear(p, fe.fc.weight, fe.fc.bias))
    pn = net.mapnet
    out = torch.cat((F.linear(feat, pn.fc_xyz.weight, pn.fc_xyz.bias), F.linear(feat, pn.fc_wpqr.weight, pn.fc_wpqr.bias)), 1)
    return out.view(n, t, 6)


def grads(net, crit, x, t):
    for p in net.parameters():
        p.grad = None
    out = forward(net, x)
    loss = crit(out, t)
    loss.backward()
    return loss.item(), out.detach(), {k: v.grad.clone() for k, v in net.mapnet.named_parameters()}


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
    l0, p0, g0 = grads(net, crit, x, t)
    print("batch %d windows x 3 = %d images %dx%d; fp32 step %.1f s; loss %.4f" % (n, n * 3, H, W, time.time() - t0, l0), flush=True)
    print("%-72s %10s %10s  %s" % ("backward arithmetic (forward exact)", "grad all", "grad worst", "(tensor)"))
    rows = [("fp16 conv operands (dY, W, X), fp32 activation gradients, scale %g" % scale, True, False, scale),
            ("fp16 conv operands + fp16 activation gradients, scale %g" % scale, True, True, scale),
            ("fp16 conv operands, fp32 activation gradients, scale 1", True, False, 1.0),
            ("fp16 conv operands, fp32 activation gradients, scale 65536", True, False, 65536.0)]
    for name, ops, gact, s in rows:
        MODE.update(ops=ops, gact=gact, scale=s)
        l, p, g = grads(net, crit, x, t)
        a, w, wn = compare(g, g0)
        print("%-72s %10.3e %10.3e  %s" % (name, a, w, wn), flush=True)


if __name__ == "__main__":
    main()
