"""What would fp8 CROSS TERMS cost the parity mode's forward pass?  ("open after round 5")

CPU only; the oracle is the instrument.  The fp16x2 forward pass contracts hi*hi + hi*lo + lo*hi on the fp16 matrix pipe (3 MFMAs per
product).  gfx950's block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4) runs at twice the fp16 rate, and a cross term is a 2^-11
correction: computed from fp8 (e4m3, one power-of-two scale per 32 channels: MXFP8) operands it would cost half an fp16 MFMA each --
2 instead of 3 MFMA-equivalents per product.  This tool replays the oracle's forward pass with every convolution computed as

    conv(hi(A), hi(W))  +  conv(q8(lo(A)), q8(hi(W)))  +  conv(q8(hi(A)), q8(lo(W)))        (fp32 accumulation)

and reports the pose deviation from the unmodified fp32 oracle (north-star bar: 1e-3 max abs) beside the exact-cross-term form
(= fp16x2) and the no-cross-term form (= fp16 operands, fp32 storage).

    python tools/fp8_cross_budget.py [windows] [H] [W]
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (tooling, not product)
from tools.fp16_budget_backward import bn_train  # noqa: E402

MODE = {"cross": "exact", "from_layer": 0, "layer": 0}  # "fp8f" from_layer = L: fp8 cross terms from layer L on, fp16 ones before


def hi_lo(x):
    hi = x.half().float()
    return hi, (x - hi).half().float()


def q8(x, dim):
    """MXFP8 e4m3: blocks of 32 along `dim` share a power-of-two scale that puts the block's maximum in [128, 256)"""
    x = x.movedim(dim, -1)
    shp = x.shape
    c = shp[-1]
    pad = (-c) % 32
    if pad:
        x = F.pad(x, (0, pad))
    b = x.reshape(*x.shape[:-1], -1, 32)
    m = b.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    s = torch.exp2(torch.floor(torch.log2(m)) - 7.0)
    q = (b / s).to(torch.float8_e4m3fn).float() * s
    q = q.reshape(*x.shape)[..., :c]
    return q.movedim(-1, dim)


def q8t(x):
    """fp8 e4m3 with ONE power-of-two scale per tensor (the maximum lands in [128, 256)): what constant scale operands would give"""
    s = torch.exp2(torch.floor(torch.log2(x.abs().amax().clamp_min(1e-30))) - 7.0)
    return (x / s).to(torch.float8_e4m3fn).float() * s


def q8f(x, e):
    """fp8 e4m3 of x * 2^e (clamped to +-448), one FIXED exponent per operand class: activations hi 2^-1, lo 2^9; weights hi 2^6, lo
    2^16 -- constants a kernel can hold in its scale operands"""
    return (x * 2.0 ** e).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * 2.0 ** -e


def conv(x, w, stride, pad):
    if MODE["cross"] == "fp32":
        return F.conv2d(x, w, None, stride, pad)
    xh, xl = hi_lo(x)
    wh, wl = hi_lo(w)
    y = F.conv2d(xh, wh, None, stride, pad)
    if MODE["cross"] == "exact" or (MODE["cross"] == "fp8f" and MODE["layer"] < MODE["from_layer"]):
        y = y + F.conv2d(xl, wh, None, stride, pad) + F.conv2d(xh, wl, None, stride, pad)
    elif MODE["cross"] == "fp8":
        y = y + F.conv2d(q8(xl, 1), q8(wh, 1), None, stride, pad) + F.conv2d(q8(xh, 1), q8(wl, 1), None, stride, pad)
    elif MODE["cross"] == "fp8f":
        y = y + F.conv2d(q8f(xl, 9), q8f(wh, 6), None, stride, pad) + F.conv2d(q8f(xh, -1), q8f(wl, 16), None, stride, pad)
    elif MODE["cross"] == "fp8t":
        y = y + F.conv2d(q8t(xl), q8t(wh), None, stride, pad) + F.conv2d(q8t(xh), q8t(wl), None, stride, pad)
    return y


def forward(net, x):
    fe = net.mapnet.feature_extractor
    n, t = x.shape[:2]
    x = x.reshape(n * t, *x.shape[2:])
    y = conv(x, fe.conv1.weight, 2, 3)
    a = F.max_pool2d(F.relu(bn_train(y, fe.bn1)), 3, 2, 1)
    for li in range(1, 5):
        MODE["layer"] = li
        for blk in getattr(fe, "layer%d" % li):
            st = blk.conv1.stride[0]
            a1 = F.relu(bn_train(conv(a, blk.conv1.weight, st, 1), blk.bn1))
            z = bn_train(conv(a1, blk.conv2.weight, 1, 1), blk.bn2)
            # what a reader of the STORED activation gets (residual add, average pool): hi + lo, with lo an fp8 in the fixed-exponent form
            stored = ((lambda t: hi_lo(t)[0] + q8f(hi_lo(t)[1], 9)) if MODE["cross"] == "fp8f" and li >= MODE["from_layer"]
                      else (lambda t: t))
            sc = bn_train(conv(a, blk.downsample[0].weight, st, 0), blk.downsample[1]) if blk.downsample is not None else stored(a)
            a = F.relu(z + sc)
    p = (stored(a) if MODE["cross"] == "fp8f" else a).mean((2, 3))
    feat = F.relu(F.linear(p, fe.fc.weight, fe.fc.bias))
    pn = net.mapnet
    out = torch.cat((F.linear(feat, pn.fc_xyz.weight, pn.fc_xyz.bias), F.linear(feat, pn.fc_wpqr.weight, pn.fc_wpqr.bias)), 1)
    return out.view(n, t, 6)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 341
    torch.manual_seed(7)
    net = oracle.MapNet(oracle.PoseNet(oracle.resnet34(), droprate=0.0, pretrained=False))
    x, _ = oracle.make_batch("mapnet", n, H, W, seed=7)
    net.train()
    with torch.no_grad():
        MODE["cross"] = "fp32"
        ref = forward(net, x)
        print("batch %d windows x 3 = %d images %dx%d, pose scale %.2f" % (n, n * 3, H, W, ref.abs().max().item()), flush=True)
        print("%-66s %10s %10s" % ("convolutions contracted as", "pose max", "pose rms"))
        for name, m, L in (("... FIXED exponents, fp8 cross terms from layer2 on (stem, layer1: fp16 cross terms)", "fp8f", 2),
                           ("... FIXED exponents, fp8 cross terms from layer3 on", "fp8f", 3),
                           ("... FIXED exponents, fp8 cross terms in layer4 only", "fp8f", 4)):
            MODE["cross"], MODE["from_layer"] = m, L
            d = forward(net, x) - ref
            print("%-66s %10.3e %10.3e" % (name, d.abs().max().item(), d.pow(2).mean().sqrt().item()), flush=True)
        MODE["from_layer"] = 0
        for name, m in (("hi*hi + hi*lo + lo*hi, all fp16 (= fp16x2: 3 MFMAs per product)", "exact"),
                        ("hi*hi fp16 + both cross terms in MXFP8 e4m3 (2 MFMA-equivalents)", "fp8"),
                        ("... with one power-of-two scale per TENSOR instead of per 32 channels", "fp8t"),
                        ("... with FIXED exponents (A: hi 2^-1, lo 2^9; W: hi 2^6, lo 2^16: the kernels')", "fp8f"),
                        ("hi*hi only (fp16 operands, fp32 tensors: 1 MFMA)", "none")):
            MODE["cross"] = m
            d = forward(net, x) - ref
            print("%-66s %10.3e %10.3e" % (name, d.abs().max().item(), d.pow(2).mean().sqrt().item()), flush=True)


if __name__ == "__main__":
    main()
