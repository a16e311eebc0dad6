"""Per-launch timing of layer1's h2 forward convolution (64 -> 64 channels, 64 x 86 pixels, B images): the register-resident-weight
kernel (csrc/halo_h2.h, mn_op_conv_halo_h2) against the chunk-resident 64-column shape (igemm_halo.h through mn_op_igemm, dtype 3).
usage: python tools/halo_h2_bench.py [B]"""
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

lib = _binding.hip()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 192
H, W = 64, 86


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


g, _, _ = checks.fwd_geom(B, H, W, 64, 64, 3, 1, 1)
x = checks.to_h2(torch.randn(B, H, W, 64, device="cuda"))
w = checks.to_h2((torch.randn(64, 3, 3, 64, device="cuda") * 0.05).reshape(64, 576))
y0 = torch.empty(B, H, W, 64, device="cuda")
y1 = torch.empty(B, H, W, 64, device="cuda")
st = torch.zeros(lib.op_igemm_grid_m(g.M), 2, 64, device="cuda")
acc = torch.zeros(256, 2, 64, device="cuda", dtype=torch.double)
one = C.c_float(1.0)
zp = checks.zero_page("cuda")
flops = 2.0 * g.M * 64 * 576
t_old = timeit(lambda: lib.op_igemm(3, C.byref(g), ptr(x), ptr(w), ptr(y0), 64, ptr(st), None, 0, None, None, one, ptr(zp), None))
t_new = timeit(lambda: lib.check(lib.op_conv_halo_h2(C.byref(g), ptr(x), ptr(w), ptr(y1), 64, ptr(acc), 256, None)))
torch.cuda.synchronize()
print("layer1 h2 forward, %d images: chunk-resident kernel %.1f us (%.0f TF fp32-equivalent), weights-in-registers kernel %.1f us (%.0f TF); "
      "max |difference| %.3e of %.3e" % (B, t_old, flops / t_old / 1e6, t_new, flops / t_new / 1e6, (y0 - y1).abs().max().item(),
                                         y0.abs().max().item()))
