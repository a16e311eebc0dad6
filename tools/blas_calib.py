"""Calibration only (not part of the product path): what the vendor GEMM reaches on this box for the
GEMM shapes the conv layers reduce to.  usage: python tools/blas_calib.py"""
import torch

def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for (M, N, K) in ((264192, 128, 1152), (67584, 256, 2304), (16896, 512, 4608), (1056768, 64, 576), (32768, 1024, 2048),
                  (8192, 8192, 8192), (4608, 512, 16896), (2304, 256, 67584)):
    A = torch.randn(M, K, device="cuda", dtype=torch.half)
    B = torch.randn(N, K, device="cuda", dtype=torch.half)
    us = t(lambda: torch.matmul(A, B.t()))
    print("M=%8d N=%5d K=%6d  %8.1f us  %7.0f TF" % (M, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)
