"""GPU micro-benchmark of the conv MFMA kernels at the BASELINE layer shapes (B=192, fp16 by default).
usage: python tools/conv_bench.py [fp16|fp32|fp32x3|fp16x2] [B]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import checks  # noqa: E402
from geomapnet_amd import _binding  # noqa: E402
from geomapnet_amd._binding import ptr  # noqa: E402

# MN_LIB: an alternative build of the library (the ablation build, `make -C geomapnet_amd/csrc ablation`)
lib = _binding.Binding(C.CDLL(os.environ["MN_LIB"])) if os.environ.get("MN_LIB") else _binding.hip()
dtype = {"fp16": 1, "fp32": 0, "fp32x3": 2, "fp16x2": 3, "fp16x2q": 5}[sys.argv[1] if len(sys.argv) > 1 else "fp16"]
H2 = dtype in (3, 5)


def up(t, weight=False):  # a channels-last operand in the storage form of `dtype` (dtype 3: h2 pairs, split on the device)
    if dtype == 5:
        return checks.to_h2q(t.cpu(), weight)[0].cuda()
    return checks.to_h2(t) if H2 else t.to(td)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 192
td = checks.TD[dtype]
one = C.c_float(1.0)


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us


def bench(name, H, W, Ci, Co, k, stride, pad):
    if os.environ.get("CB_MATCH") and os.environ["CB_MATCH"] not in name:
        return
    g, Ho, Wo = checks.fwd_geom(B, H, W, Ci, Co, k, stride, pad)
    gd, _, _ = checks.dgrad_geom(B, H, W, Ci, Co, k, stride, pad)
    w32 = torch.randn(Co, k, k, Ci, device="cuda") * 0.05
    x = up(torch.randn(B, H, W, Ci, device="cuda"))
    w = up(w32, True)
    wt = up(w32.permute(3, 1, 2, 0).contiguous(), True)
    gy = up(torch.randn(B, Ho, Wo, Co, device="cuda"))
    y = torch.empty(B, Ho, Wo, Co, dtype=td, device="cuda")
    gx = torch.empty(B, H, W, Ci, dtype=td, device="cuda")
    res, gate = torch.randn_like(gx), up(torch.randn(B, H, W, Ci, device="cuda"))
    gw = torch.zeros(Co, k * k * Ci, device="cuda")
    st = torch.zeros(lib.op_igemm_grid_m(g.M), 2, Co, device="cuda")
    flops = 2.0 * g.M * Co * k * k * Ci
    t_f = timeit(lambda: lib.op_igemm(dtype, C.byref(g), ptr(x), ptr(w), ptr(y), Co, ptr(st), None, 0, None, None, one, ptr(checks.zero_page("cuda")), None))
    t_d = timeit(lambda: lib.op_igemm(dtype, C.byref(gd), ptr(gy), ptr(wt), ptr(gx), Ci, None, None, 0, None, None, one, ptr(checks.zero_page("cuda")), None))
    t_r = timeit(lambda: lib.op_igemm(dtype, C.byref(gd), ptr(gy), ptr(wt), ptr(gx), Ci, None, None, 0, ptr(res), ptr(gate), one, ptr(checks.zero_page("cuda")), None))
    t_w = timeit(lambda: lib.op_wgrad(dtype, C.byref(g), ptr(gy), Co, ptr(x), ptr(gw), k * k * Ci, None, one, 1024, ptr(checks.zero_page("cuda")), None))
    if dtype in (1, 2, 3) and k == 3 and stride == 1:  # the plan's form: partial tiles through a workspace + reduce launch
        t_ws = timeit(lambda: lib.op_wgrad_ws(dtype, C.byref(g), ptr(gy), Co, ptr(x), ptr(gw), k * k * Ci, one, ptr(WS), WS.numel(), ptr(checks.zero_page("cuda")), None))
        print("%-22s wgrad through the workspace %7.1f us %6.0f TF" % (name, t_ws, flops / t_ws / 1e6), flush=True)
    if dtype == 1 and Ci == 64 and k == 3 and stride == 1:
        if Co == 64:  # persistent two-group form (halo_pp.h); CB_PP_WGS = workgroup counts to try
            acc = torch.zeros(8, 2, 64, dtype=torch.double, device="cuda")
            for wgs in [int(v) for v in os.environ.get("CB_PP_WGS", "0").split(",")]:
                tp_f = timeit(lambda: lib.op_conv_halo_pp(C.byref(g), ptr(x), ptr(w), ptr(y), Co, ptr(acc), 8, 0, None, None, None, one, wgs, None))
                tp_d = timeit(lambda: lib.op_conv_halo_pp(C.byref(gd), ptr(gy), ptr(wt), ptr(gx), Ci, None, 0, 0, None, None, None, one, wgs, None))
                tp_r = timeit(lambda: lib.op_conv_halo_pp(C.byref(gd), ptr(gy), ptr(wt), ptr(gx), Ci, None, 0, 0, ptr(res), None, ptr(gate), one, wgs, None))
                print("%-22s halo_pp wgs=%d: fwd+stats %7.1f us %6.0f TF | dgrad %7.1f us | +res+out_gate %7.1f us"
                      % (name, wgs, tp_f, flops / tp_f / 1e6, tp_d, tp_r), flush=True)
    io = (x.numel() + y.numel()) * x.element_size()
    print("%-22s M=%8d N=%4d K=%5d  fwd %7.1f us %6.0f TF (io %5.2f TB/s) | dgrad %7.1f us %6.0f TF | +res %7.1f us | wgrad %7.1f us %6.0f TF"
          % (name, g.M, Co, k * k * Ci, t_f, flops / t_f / 1e6, io / t_f / 1e6, t_d, flops / t_d / 1e6, t_r, t_w, flops / t_w / 1e6), flush=True)


print("dtype", {1: "fp16", 0: "fp32", 2: "fp32x3 (fp32 tensors, f16x3 igemm / bf16x3 wgrad)", 3: "fp16x2 (h2 operands, fp32 outputs)"}[dtype], "B", B)
WS = torch.empty(int(lib.op_wgrad_ws_floats()), device="cuda")
ONLY_GEMM = os.environ.get("CB_ONLY") == "gemm"
if ONLY_GEMM:
    def bench(*a):  # noqa: F811
        pass
bench("layer1 3x3 64->64", 64, 86, 64, 64, 3, 1, 1)
bench("layer2.0 3x3/2 64->128", 64, 86, 64, 128, 3, 2, 1)
bench("layer2 3x3 128->128", 32, 43, 128, 128, 3, 1, 1)
bench("layer2.0 1x1/2 64->128", 64, 86, 64, 128, 1, 2, 0)
bench("layer3 3x3 256->256", 16, 22, 256, 256, 3, 1, 1)
bench("layer4 3x3 512->512", 8, 11, 512, 512, 3, 1, 1)
if H2:
    sys.exit(0)
# stem through the pixel-pair formulation
if ONLY_GEMM:
    B = 2
g, Hp, Wp, H0, W0 = checks.stem_geom(B, 256, 341)
xp = torch.randn(B, Hp, Wp, 4, device="cuda").to(td)
wc = (torch.randn(64, 224, device="cuda") * 0.05).to(td)
y = torch.empty(B, H0, W0, 64, dtype=td, device="cuda")
st = torch.zeros(lib.op_igemm_grid_m(g.M), 2, 64, device="cuda")
t_f = timeit(lambda: lib.op_igemm(dtype, C.byref(g), ptr(xp), ptr(wc), ptr(y), 64, ptr(st), None, 0, None, None, one, ptr(checks.zero_page("cuda")), None))
if dtype == 1:
    t_s = timeit(lambda: lib.op_stem_conv(ptr(xp), ptr(wc), ptr(y), None, 0, B, 256, 341, Wp, None))
    acc = torch.zeros(8, 2, 64, dtype=torch.float64, device="cuda")
    t_s2 = timeit(lambda: lib.op_stem_conv(ptr(xp), ptr(wc), ptr(y), ptr(acc), 8, B, 256, 341, Wp, None))
    print("stem kernel (stem.h): fwd %7.1f us %6.0f TF(real) out %5.2f TB/s | with BatchNorm sums %7.1f us" % (t_s, 2.0 * g.M * 64 * 147 / t_s / 1e6, y.numel() * 2 / t_s / 1e6, t_s2))
gy = torch.randn_like(y)
if dtype == 1:  # stem backward: the two-launch form (stem_bwd.h) vs the four launches it replaces
    Po, Qo = (H0 - 1) // 2 + 1, (W0 - 1) // 2 + 1
    a0 = torch.relu(y)
    p0 = torch.empty(B, Po, Qo, 64, dtype=td, device="cuda")
    idx = torch.empty(B, Po, Qo, 64, dtype=torch.uint8, device="cuda")
    lib.op_maxpool_fwd(1, ptr(a0), ptr(p0), ptr(idx), B, H0, W0, 64, None)
    gp = torch.randn_like(p0)
    gam, bet, mu, isd = (torch.ones(64, device="cuda"), torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda"),
                         torch.ones(64, device="cuda"))
    dg, db, coef = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda"), torch.zeros(256, device="cuda")
    acc2 = torch.zeros(128, dtype=torch.float64, device="cuda")
    gwf = torch.zeros(64, 147, device="cuda")
    cmf = torch.arange(224, dtype=torch.int32, device="cuda") % 147
    t_b = timeit(lambda: lib.op_stem_bwd(ptr(y), ptr(idx), ptr(gp), ptr(gam), ptr(bet), ptr(mu), ptr(isd), ptr(xp), ptr(gwf), 147,
                                         ptr(cmf), ptr(dg), ptr(db), ptr(coef), ptr(acc2), B, 256, 341, Wp, one, None))
    ga0 = torch.empty_like(y)
    gyy = torch.empty_like(y)
    def chain():
        lib.op_maxpool_bwd(1, ptr(idx), ptr(gp), ptr(ga0), B, H0, W0, 64, None)
        lib.op_bn_bwd(1, ptr(ga0), ptr(a0), ptr(y), B * H0 * W0, 64, ptr(gam), ptr(mu), ptr(isd), ptr(dg), ptr(db), ptr(gyy), ptr(coef),
                      ptr(acc2), one, None)
        lib.op_wgrad(1, C.byref(g), ptr(gyy), 64, ptr(xp), ptr(gwf), 147, ptr(cmf), one, 1024, ptr(checks.zero_page("cuda")), None)
    t_c = timeit(chain)
    print("stem backward: two launches (stem_bwd.h) %7.1f us | maxpool_bwd + bn_bwd + wgrad %7.1f us" % (t_b, t_c))
gw = torch.zeros(64, 147, device="cuda")
cm = torch.arange(224, dtype=torch.int32, device="cuda") % 147
t_w = timeit(lambda: lib.op_wgrad(dtype, C.byref(g), ptr(gy), 64, ptr(xp), ptr(gw), 147, ptr(cm), one, 1024, ptr(checks.zero_page("cuda")), None))
fl = 2.0 * g.M * 64 * 147
print("stem 7x7/2 3->64        M=%8d  fwd %7.1f us %6.0f TF(real) out %5.2f TB/s | wgrad %7.1f us %6.0f TF(real)"
      % (g.M, t_f, fl / t_f / 1e6, y.numel() * 2 / t_f / 1e6, t_w, fl / t_w / 1e6))

# asymptotic GEMM rate of the same kernel: 1x1 conv with a long K loop (no taps, 64+ K-steps)
for (Ci, Co, Bq) in ((2048, 1024, 32), (4096, 512, 32), (2304, 256, 66), (8192, 8192, 8)):
    Hq = 32
    g, Ho, Wo = checks.fwd_geom(Bq, Hq, Hq, Ci, Co, 1, 1, 0)
    x = torch.randn(Bq, Hq, Hq, Ci, device="cuda").to(td)
    w = (torch.randn(Co, Ci, device="cuda") * 0.02).to(td)
    y = torch.empty(Bq, Hq, Hq, Co, dtype=td, device="cuda")
    t = timeit(lambda: lib.op_igemm(dtype, C.byref(g), ptr(x), ptr(w), ptr(y), Co, None, None, 0, None, None, one, ptr(checks.zero_page("cuda")), None))
    print("plain GEMM M=%d N=%d K=%d: %7.1f us %6.0f TF" % (g.M, Co, Ci, t, 2.0 * g.M * Co * Ci / t / 1e6))
