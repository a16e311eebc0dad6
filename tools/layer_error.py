"""Per-layer error of the HIP path against the oracle on the benchmark batch (GPU tool; the oracle is the instrument).

    python tools/layer_error.py [windows] [H] [W]        (default 64 256 341: BASELINE configs[2])

One training step of the HIP fp32 build and of the HIP fp16 build on the oracle's weights and batch; every stored
activation of the forward pass (stem conv output, pooled stem output, every residual block's output, the feature vector,
the poses) and the gradient arriving at every block's output are read back through mn_debug_tensor and compared with the
oracle's tensors: relative L2 error and max abs error per tensor.  The fp16 column is the per-layer budget of the
deviation `bench.py` reports in `parity` (tools/fp16_budget.py reproduces the same growth on the CPU by rounding the
oracle's own tensors at the same storage points).
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (tooling, not product)
import geomapnet_amd as G  # noqa: E402

EMU = os.environ.get("MN_TOOL_EMU") == "1"  # dry-run of this script on the CPU emulator build (tiny sizes)
KW = {}
if EMU:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_lib  # noqa: E402
    KW = {"_binding": emu_lib.load()}
DEV = "cpu" if EMU else "cuda"


def oracle_tensors(net, crit, x, t):
    """forward + backward of the oracle with every tensor of interest kept (NCHW, fp32)"""
    fe = net.mapnet.feature_extractor
    keep, grads = {}, {}

    def want_grad(name, tensor):
        tensor.register_hook(lambda g, n=name: grads.__setitem__(n, g.detach()))

    n, T = x.shape[:2]
    xx = x.reshape(n * T, *x.shape[2:])
    y = fe.conv1(xx)
    keep["stem.y"] = y.detach()
    a = fe.maxpool(F.relu(fe.bn1(y)))
    keep["p0"] = a.detach()
    bi = 0
    for li in range(1, 5):
        for blk in getattr(fe, "layer%d" % li):
            y1 = blk.conv1(a)
            a1 = F.relu(blk.bn1(y1))
            y2 = blk.conv2(a1)
            z = blk.bn2(y2)
            sc = a if blk.downsample is None else blk.downsample(a)
            pre = z + sc
            want_grad("b%d.gout" % bi, pre)  # the HIP path stores the gradient at a block's output already ReLU-gated
            a = F.relu(pre)
            keep["b%d.y1" % bi] = y1.detach()
            keep["b%d.out" % bi] = a.detach()
            bi += 1
    p = fe.avgpool(a).flatten(1)
    feat = F.relu(fe.fc(p))
    keep["feat"] = feat.detach()
    pn = net.mapnet
    poses = torch.cat((pn.fc_xyz(feat), pn.fc_wpqr(feat)), 1)
    keep["poses"] = poses.detach()
    loss = crit(poses.view(n, T, 6), t)
    loss.backward()
    return keep, grads, loss.item()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 341
    torch.manual_seed(7)
    onet = oracle.MapNet(oracle.PoseNet(oracle.resnet34(), droprate=0.0, pretrained=False))
    ocrit = oracle.MapNetCriterion(0.0, -3.0, 0.0, -3.0, True, True)
    sd0 = {k: v.clone() for k, v in onet.state_dict().items()}
    x, t = oracle.make_batch("mapnet", n, H, W, seed=7)
    onet.train()
    keep, ograds, lo = oracle_tensors(onet, ocrit, x, t)
    names = list(keep.keys())
    table = {}
    for dtype in ("fp32", "fp16"):
        G.set_compute_dtype(dtype)
        net = G.MapNet(G.PoseNet(G.resnet34(**KW), droprate=0.0, pretrained=False, **KW))
        net.load_state_dict(sd0)
        crit = G.MapNetCriterion(sax=0.0, saq=-3.0, srx=0.0, srq=-3.0, learn_beta=True, learn_gamma=True, **KW)
        if not EMU:
            net.cuda()
            crit.cuda()
        opt = G.Optimizer([{"params": net.parameters()}, {"params": [crit.sax, crit.saq]}, {"params": [crit.srx, crit.srq]}],
                          "adam", base_lr=0.0, weight_decay=0.0)
        net.train()
        l, _ = G.step_feedfwd(x.to(DEV), net, not EMU, t.to(DEV), crit, opt, True)
        if not EMU:
            torch.cuda.synchronize()
        eng = net.mapnet._engine
        plan = next(iter(eng.plans.values()))
        scale = eng.loss_scale_state()[0] if dtype == "fp16" else 1.0

        def cmp(name, ref, mul=1.0):
            got = eng.debug_tensor(plan, name).float().cpu() * mul
            if ref.dim() == 4:
                got = got.view(ref.shape[0], ref.shape[2], ref.shape[3], ref.shape[1]).permute(0, 3, 1, 2)
            else:
                got = got.view(ref.shape)
            d = (got - ref).double()
            table.setdefault(name, {})[dtype] = (float(d.norm() / (ref.double().norm() + 1e-30)), float(d.abs().max()),
                                                 float(ref.abs().max()))

        for name in names:
            cmp(name, keep[name])
        for name, g in ograds.items():
            cmp(name, g, 1.0 / scale)
        table.setdefault("loss", {})[dtype] = (abs(l - lo) / max(1.0, abs(lo)), abs(l - lo), abs(lo))
        del net, crit, opt
    print("%d windows x 3 = %d images %dx%d; oracle loss %.6f" % (n, n * 3, H, W, lo))
    print("%-12s | %-34s | %-34s | %s" % ("tensor", "HIP fp32: rel L2   max |d|", "HIP fp16: rel L2   max |d|", "max |ref|"))
    for name in ["loss"] + names + sorted(ograds, key=lambda s: -int(s[1:s.index(".")])):
        a, b = table[name]["fp32"], table[name]["fp16"]
        print("%-12s | %14.3e %14.3e      | %14.3e %14.3e      | %.3e" % (name, a[0], a[1], b[0], b[1], a[2]))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "layer_error.json"), "w") as f:
            json.dump({"windows": n, "H": H, "W": W, "table": table}, f)


if __name__ == "__main__":
    main()
