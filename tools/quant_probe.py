"""Tile-quantisation probe: the layer3 conv (256->256, 3x3) at row counts just below / above whole rounds of the
512 resident tile slots.  usage: python tools/quant_probe.py"""
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
    return a.elapsed_time(b) / n * 1e3


SHAPES = ((256, 256, 16, 32, (64, 120, 128, 132, 136, 192, 248, 256, 264)),
          (512, 512, 8, 16, (120, 128, 132, 136, 256, 264)),
          (128, 128, 32, 32, (120, 128, 129, 136, 256, 258)))
if len(sys.argv) > 1 and sys.argv[1] == "layers":  # the three real row counts only
    SHAPES = ((256, 256, 16, 32, (128, 132)), (512, 512, 8, 16, (128, 132)), (128, 128, 32, 32, (256, 258)))
for (Ci, Co, H, W, Bs) in SHAPES:
    for B in Bs:
        g, Ho, Wo = checks.fwd_geom(B, H, W, Ci, Co, 3, 1, 1)
        x = torch.randn(B, H, W, Ci, device="cuda").half()
        w = (torch.randn(Co, 3, 3, Ci, device="cuda") * 0.05).half()
        y = torch.empty(B, Ho, Wo, Co, dtype=torch.half, device="cuda")
        st = torch.zeros(lib.op_igemm_grid_m(g.M), 2, Co, device="cuda")
        if os.environ.get("SK"):
            ws = torch.zeros(512, 2, 128 * 128, device="cuda")
            cnt = torch.zeros(512, dtype=torch.int32, device="cuda")
            t = timeit(lambda: lib.op_igemm_streamk(1, C.byref(g), ptr(x), ptr(w), ptr(y), Co, ptr(st), None, 0, None, None, one,
                                                    ptr(ws), ptr(cnt), 512, None))
        else:
          t = timeit(lambda: lib.op_igemm(1, C.byref(g), ptr(x), ptr(w), ptr(y), Co, ptr(st), None, 0, None, None, one,
                                         ptr(checks.zero_page("cuda")), None))
        tiles = ((g.M + 127) // 128) * (Co // 128)
        print("C=%d M=%7d tiles128=%5d (%.2f rounds of 512)  %7.1f us  %5.0f TF  %6.3f us/tile-slot-round"
              % (Ci, g.M, tiles, tiles / 512.0, t, 2.0 * g.M * Co * 9 * Ci / t / 1e6, t / max(1, -(-tiles // 512))), flush=True)
