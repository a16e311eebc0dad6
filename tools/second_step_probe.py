"""GPU: how far is the HIP path's SECOND training step from the oracle's, under which knobs?  (diagnostic for
tests/test_gpu_parity.py::test_second_step_fp32_with_a_smooth_update)
usage: python tools/second_step_probe.py [fp32|fp16] [lr] [adam_eps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import checks  # noqa: E402
from geomapnet_amd import _binding  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
eps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
lib = _binding.hip()
rep = checks.check_train_step(lib, "cuda", dtype, mode="mapnet", N=2, H=64, W=85, steps=3, lr=lr, adam_eps=eps, loss_rtol=1.0,
                              pose_atol=100.0, grad_l2_rtol=None)
knobs = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("MN_"))
for i, row in enumerate(rep):
    if row[0] == "displacement_rel_l2":
        print("[%s] displacement_rel_l2 %.3e" % (knobs, row[1]))
    else:
        l, lo, pe = row
        print("[%s] step %d loss %.6f oracle %.6f rel %.2e pose max err %.2e" % (knobs, i + 1, l, lo, abs(l - lo) / max(1, abs(lo)), pe))
