"""Consumer-side BatchNorm fusion, measured (VERDICT rounds 2-4): conv3x3 of layers 3 / 4 at the benchmark batch with the
BatchNorm apply + ReLU of the producing layer applied to the A fragments inside the convolution (igemm_halo.h FBN) against the
same convolution reading the stored activation.  The fusion would replace one bn_apply launch per convolution (its time at the
same tensor size: the step's serial profile by grid, printed by tools/prof_mode.sh) -- and only if the weight gradient of the
same convolution took its X operand through the same transform.
usage: python tools/fbn_bench.py [B]"""
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
one = C.c_float(1.0)


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us


def bench(name, H, W, Cc):
    g, Ho, Wo = checks.fwd_geom(B, H, W, Cc, Cc, 3, 1, 1)
    y = torch.randn(B, H, W, Cc, device="cuda").half()
    scale = 0.5 + torch.rand(Cc, device="cuda")
    shift = torch.randn(Cc, device="cuda") * 0.3
    a1 = torch.relu(y.float() * scale + shift).half()
    w = (torch.randn(Cc, 3, 3, Cc, device="cuda") * 0.05).half()
    coef = torch.cat((scale, shift)).contiguous()
    out0 = torch.empty(B, Ho, Wo, Cc, dtype=torch.float16, device="cuda")
    out1 = torch.empty_like(out0)
    zp = checks.zero_page("cuda")
    plain = lambda: lib.check(lib.op_igemm(1, C.byref(g), ptr(a1), ptr(w), ptr(out0), Cc, None, None, 0, None, None, one, ptr(zp), None))  # noqa: E731
    fused = lambda: lib.check(lib.op_igemm_fbn(C.byref(g), ptr(y), ptr(coef), ptr(w), ptr(out1), Cc, None))  # noqa: E731
    res = []
    for rep in range(3):
        res.append((timeit(plain), timeit(fused)))
    torch.cuda.synchronize()
    err = (out1.float() - out0.float()).abs().max().item() / out0.float().abs().max().item()
    tp, tf = min(r[0] for r in res), min(r[1] for r in res)
    # an elementwise pass over the same tensor (read 2 B, write 2 B per element) as the stand-in for the bn_apply launch's floor
    ew = timeit(lambda: torch.relu_(a1))
    print("%-20s conv on the stored activation %6.1f us | BatchNorm apply + ReLU fused into the operand path %6.1f us (%+5.1f us) | "
          "an in-place elementwise pass over that activation %5.1f us | fused vs unfused outputs: max rel diff %.1e"
          % (name, tp, tf, tf - tp, ew, err), flush=True)


print("B", B)
bench("layer3 256->256", 16, 22, 256)
bench("layer4 512->512", 8, 11, 512)
