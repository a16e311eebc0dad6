"""Where does the fp16 build's pose deviation come from?  (CPU only; the oracle is the instrument.)

The HIP fp16 path stores the input image, every conv operand (weights and activations) and every activation in fp16 and
accumulates in fp32.  This tool replays the oracle's forward pass (training-mode BatchNorm, the benchmark batch) with fp16
ROUNDING inserted at selectable storage points and reports the deviation of the predicted poses from the unrounded fp32
forward -- a per-storage-class and per-stage error budget that needs no GPU:

    python tools/fp16_budget.py [windows] [H] [W]

classes:  I = input image, W = conv weights, Y = raw conv outputs (pre-BatchNorm), A = post-activation tensors
(a1, block outputs, pooled stem output);  stages: stem, layer1..layer4.
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (tooling, not product)


def r16(t, on):
    return t.half().float() if on else t


def bn_train(x, bn):
    return F.batch_norm(x, None, None, bn.weight, bn.bias, True, 0.0, bn.eps)


def forward(net, x, cls, stages):
    """cls: set of storage classes to round; stages: set of stage names where they apply"""
    fe = net.mapnet.feature_extractor
    n, t = x.shape[:2]
    x = x.reshape(n * t, *x.shape[2:])

    def on(c, st):
        return c in cls and st in stages

    x = r16(x, on("I", "stem"))
    y = r16(F.conv2d(x, r16(fe.conv1.weight, on("W", "stem")), None, 2, 3), on("Y", "stem"))
    a = F.relu(bn_train(y, fe.bn1))
    a = r16(F.max_pool2d(r16(a, on("A", "stem")), 3, 2, 1), on("A", "stem"))
    for li in range(1, 5):
        st = "layer%d" % li
        for blk in getattr(fe, st):
            y1 = r16(F.conv2d(a, r16(blk.conv1.weight, on("W", st)), None, blk.conv1.stride, 1), on("Y", st))
            a1 = r16(F.relu(bn_train(y1, blk.bn1)), on("A", st))
            y2 = r16(F.conv2d(a1, r16(blk.conv2.weight, on("W", st)), None, 1, 1), on("Y", st))
            z = bn_train(y2, blk.bn2)
            if blk.downsample is not None:
                yd = r16(F.conv2d(a, r16(blk.downsample[0].weight, on("W", st)), None, blk.downsample[0].stride, 0), on("Y", st))
                sc = r16(bn_train(yd, blk.downsample[1]), on("A", st))
            else:
                sc = a
            a = r16(F.relu(z + sc), on("A", st))
    p = a.mean((2, 3))
    feat = F.relu(F.linear(p, fe.fc.weight, fe.fc.bias))
    pn = net.mapnet
    return torch.cat((F.linear(feat, pn.fc_xyz.weight, pn.fc_xyz.bias), F.linear(feat, pn.fc_wpqr.weight, pn.fc_wpqr.bias)), 1)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 341
    torch.manual_seed(7)
    net = oracle.MapNet(oracle.PoseNet(oracle.resnet34(), droprate=0.0, pretrained=False))
    x, _ = oracle.make_batch("mapnet", n, H, W, seed=7)
    ALL = {"stem", "layer1", "layer2", "layer3", "layer4"}
    with torch.no_grad():
        t0 = time.time()
        ref = forward(net, x, set(), set())
        print("batch %d windows x 3 = %d images %dx%d; fp32 forward %.1f s; |pose| max %.3f rms %.3f"
              % (n, n * 3, H, W, time.time() - t0, ref.abs().max(), ref.pow(2).mean().sqrt()), flush=True)
        rows = [("everything the HIP fp16 build rounds (I W Y A, all stages)", set("IWYA"), ALL),
                ("I   input image only", set("I"), ALL), ("W   conv weights only", set("W"), ALL),
                ("Y   raw conv outputs only", set("Y"), ALL), ("A   post-activation tensors only", set("A"), ALL)]
        rows += [("WYA %s only" % st, set("IWYA"), {st}) for st in ("stem", "layer1", "layer2", "layer3", "layer4")]
        rows += [("WYA all but %s" % st, set("IWYA"), ALL - {st}) for st in ("stem", "layer1", "layer4")]
        rows += [("WYA layers 2-4 only (stem + layer1 kept fp32)", set("IWYA"), {"layer2", "layer3", "layer4"})]
        print("%-62s %12s %12s" % ("fp16 rounding at", "pose max|d|", "pose rms d"))
        for name, cls, stages in rows:
            out = forward(net, x, cls, stages)
            d = out - ref
            print("%-62s %12.3e %12.3e" % (name, d.abs().max(), d.pow(2).mean().sqrt()), flush=True)


if __name__ == "__main__":
    main()
