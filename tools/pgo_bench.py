"""Pose-graph optimisation throughput: mn_pgo_optimize (one wavefront per window, fp64) on the GPU vs the CPU oracle
(numpy/scipy restatement of the reference's PoseGraph, oracle/pgo.py) on a bounded sample.
  python tools/pgo_bench.py [W]
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import checks  # noqa: E402
from geomapnet_amd import _binding  # noqa: E402
from oracle import pgo as opgo  # noqa: E402


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    lib = _binding.hip()
    for fc in (False, True):
        pred, vos, _ = checks.pgo_windows(W, 7, fc, seed=1)
        d_pred, d_vos = torch.from_numpy(pred).cuda(), torch.from_numpy(vos).cuda()
        d_out = torch.empty_like(d_pred)
        d_st = torch.empty(W, dtype=torch.int32, device="cuda")
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

        def run():
            lib.check(lib.pgo_optimize(_binding.ptr(d_pred), _binding.ptr(d_vos), _binding.ptr(d_out), _binding.ptr(d_st), W, 7,
                                       int(fc), 1.0, 1.0, 1.0, 1.0, 10, stream))
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        n_cpu = 32
        t0 = time.perf_counter()
        ref = [opgo.optimize_window(pred[w], vos[w], fc=fc) for w in range(n_cpu)]
        cpu_s = (time.perf_counter() - t0) / n_cpu
        err = np.abs(d_out[:n_cpu].cpu().numpy() - np.stack(ref)).max()
        print("pgo N=7 %-5s W=%d: %.3f ms/launch = %.2f Mwindows/s | CPU oracle %.2f ms/window (%d windows, 1 core) -> x%.0f | max err %.1e"
              % ("fc" if fc else "chain", W, ms, W / ms / 1e3, cpu_s * 1e3, n_cpu, cpu_s * 1e3 * W / ms, err))


if __name__ == "__main__":
    main()
