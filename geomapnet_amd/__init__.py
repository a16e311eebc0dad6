"""geomapnet_amd: the MapNet training hot path of NVlabs/geomapnet as hand-written gfx950
(MI355X) kernels behind the reference's own Python API.  See DESIGN.md / INTEGRATION.md."""
from .engine import set_compute_dtype, get_compute_dtype  # noqa: F401
from .posenet import PoseNet, MapNet, resnet34  # noqa: F401
from .criterion import PoseNetCriterion, MapNetCriterion, MapNetOnlineCriterion  # noqa: F401
from .optimizer import Optimizer  # noqa: F401
from .train import step_feedfwd, load_state_dict, save_checkpoint, load_checkpoint  # noqa: F401
from .feed import DeviceFeed  # noqa: F401  (batch k+1's host-to-device copy under step k)
from . import evaluate  # noqa: F401  (scripts/eval.py flow + error metric)
from . import data  # noqa: F401  (MF / MFOnline batch construction, VO targets, process_poses)
