"""CPU oracle for the MapNet training hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This package restates, in plain CPU PyTorch (fp32 or fp64), the algorithm of the reference
hot path (NVlabs/geomapnet): ResNet-34 feature extractor + PoseNet/MapNet heads, the three
pose criteria, the log-quaternion relative-pose math, and one `step_feedfwd` training step
with Adam.  Every function cites the reference file:line it follows.

Rules (enforced by tests/test_host_logic.py::test_product_never_imports_the_oracle_or_the_emulator):
  * only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import it;
  * nothing under geomapnet_amd/ imports it -- the product path is the HIP library and fails
    loudly when that library is missing.

Pinning: the restatement is checked (a) against the reference's own modules executed from
/root/reference where that tree exists (tests/test_oracle_pinned_to_reference.py; the tree
does not travel to the GPU box) and (b) everywhere against golden vectors generated from those
reference modules and committed under tests/golden/ (generator: tests/golden/make_golden.py).
The reference ships no golden vectors or asserting tests of its own for this path
(SURVEY.md section 4), so executing the reference modules is what pins parity.

Third-party arithmetic the reference delegates to and that is absent from /root/reference:
torchvision.models.resnet34 (version unpinned in environment.yml:15; pairs with
pytorch 0.4.1 => torchvision 0.2.x) -- restated in oracle/resnet34.py from its published
architecture (He et al. 2015, BasicBlock [3,4,6,3]); torch.optim.Adam and
torch.nn.utils.clip_grad_norm (PyTorch 0.4.1) -- restated in oracle/train_step.py.
"""
from .resnet34 import resnet34, ResNet34  # noqa: F401
from .posenet import PoseNet, MapNet  # noqa: F401
from .criterion import PoseNetCriterion, MapNetCriterion, MapNetOnlineCriterion  # noqa: F401
from . import pose_math  # noqa: F401
from .train_step import Optimizer, step_feedfwd, adam_reference_step  # noqa: F401
from .synthetic import make_batch  # noqa: F401
