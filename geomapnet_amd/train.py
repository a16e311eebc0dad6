"""`step_feedfwd`: one training / validation step, API of /root/reference/common/train.py:322-363.

Train branch = ONE fused library call (mn_train_step): H2D copy aside, forward, criterion,
zero_grad, backward, optional clip_grad_norm, Adam -- then `loss.item()` (the same blocking
read-back the reference does at :361).  With torch.distributed initialised (world > 1) the step
is issued in stages so each gradient bucket's RCCL all-reduce overlaps the remaining backward
(geomapnet_amd/dp.py).
"""
import os

import torch

from .criterion import _Criterion
from .engine import MODE_POSENET
from .posenet import MapNet, engine_of
from . import dp


# diagnostic: issue the step in the staged (data-parallel) form even on one GPU, e.g. to measure its overhead
_FORCE_STAGED = os.environ.get("MN_FORCE_STAGED", "0") == "1"


def _bind(engine, criterion, optim):
    store = engine.crit_slice()
    if criterion._store.data_ptr() != store.data_ptr():
        criterion._rebind(store)
    if optim is not None and getattr(optim.learner, "_engine", None) is not engine:
        optim.learner._attach(engine)


def step_feedfwd(data, model, cuda, target=None, criterion=None, optim=None, train=True, max_grad_norm=0.0):
    if train:
        assert criterion is not None
    engine = engine_of(model)
    dev = engine.device
    data = data.to(dev, non_blocking=True) if data.device != dev else data
    if not train or criterion is None:
        output = model(data)  # in the model's current mode, as the reference (its caller sets model.eval())
        if criterion is None:
            return 0, output
        loss = criterion(output, target.to(dev, non_blocking=True))
        return loss.item(), output

    if not isinstance(criterion, _Criterion):
        raise TypeError("step_feedfwd(train=True) needs a geomapnet_amd criterion")
    if not model.training:
        raise RuntimeError("step_feedfwd(train=True) on a model in eval mode")
    target = target.to(dev, non_blocking=True).float().contiguous()
    data = data.contiguous() if engine.input_u8 is not None else data.float().contiguous()
    mode = criterion.mode
    if mode == MODE_POSENET:
        if data.dim() != 4:
            raise ValueError("PoseNet training expects data [N,3,H,W]")
        n, t = data.shape[0], 1
    else:
        if data.dim() != 5 or not isinstance(model, MapNet):
            raise ValueError("MapNet training expects a MapNet model and data [N,T,3,H,W]")
        n = data.shape[0]
        t = data.shape[1] if mode == 1 else data.shape[1] // 2
    H, W = engine.image_dims(data)
    # the fused kernel indexes the target as [n][rows of this criterion][6]: a wrong layout must not reach the device
    criterion.check_batch(n, data.shape[1] if mode != MODE_POSENET else 1, target)
    _bind(engine, criterion, optim)
    plan = engine.plan(mode, n, t, H, W)
    lr, wd, betas, eps = optim.learner.hyper()
    engine.configure_step(plan, lr, wd, betas, eps, float(max_grad_norm), criterion.learn_beta, criterion.learn_gamma,
                          method=optim.learner.method())
    if dp.world_size() > 1 or _FORCE_STAGED:
        loss, poses = dp.train_step(engine, plan, data, target)
    else:
        loss, poses = engine.train_step(plan, data, target)
    output = poses.view(n, -1, 6) if mode != MODE_POSENET else poses
    return loss.item(), output


# ---- checkpoints: /root/reference/common/train.py:22-53 (load_state_dict), :162-178 (resume), :198-204 (save) ----
def _name_prefix(longer, shorter):
    """`longer` == prefix + `shorter` -> prefix, else None"""
    return longer[: len(longer) - len(shorter)] if longer.endswith(shorter) else None


def load_state_dict(model, state_dict):
    """model.load_state_dict for checkpoints whose parameter names differ from the model's by a leading module path --
    a PoseNet checkpoint into MapNet (`mapnet.` prefix) or the reverse (contract of common/train.py:22-53: the relation is
    read off the FIRST parameter name of each side; names that are not related by a prefix raise KeyError)."""
    model_first = next(iter(model.named_parameters()))[0]
    state_first = next(iter(state_dict.keys()))
    add = _name_prefix(model_first, state_first)       # the model's names carry an extra prefix
    strip = _name_prefix(state_first, model_first)     # the checkpoint's names do
    if add is None and strip is None:
        raise KeyError("Could not find the correct prefixes between %s and %s" % (model_first, state_first))
    if add is not None:
        renamed = ((add + k, v) for k, v in state_dict.items())
    else:
        renamed = ((k[len(strip):] if k.startswith(strip) else k, v) for k, v in state_dict.items())
    from collections import OrderedDict
    model.load_state_dict(OrderedDict(renamed))


def save_checkpoint(filename, epoch, model, optimizer, criterion):
    """the reference's checkpoint dict (common/train.py:198-204); `filename` None returns the dict only"""
    checkpoint_dict = {"epoch": epoch, "model_state_dict": model.state_dict(),
                       "optim_state_dict": optimizer.learner.state_dict(),
                       "criterion_state_dict": criterion.state_dict()}
    if filename is not None:
        torch.save(checkpoint_dict, filename)
    return checkpoint_dict


def load_checkpoint(checkpoint, model, optimizer=None, criterion=None, resume_optim=False):
    """Trainer.__init__'s resume logic (common/train.py:162-178); `checkpoint` is a dict or a file name.
    Returns the epoch to start from (0 unless resume_optim)."""
    if not isinstance(checkpoint, dict):
        checkpoint = torch.load(checkpoint, map_location=lambda storage, loc: storage, weights_only=False)
    load_state_dict(model, checkpoint["model_state_dict"])
    start_epoch = 0
    if resume_optim:
        optimizer.learner.load_state_dict(checkpoint["optim_state_dict"])
        start_epoch = checkpoint["epoch"]
        if "criterion_state_dict" in checkpoint:
            c_state = dict(checkpoint["criterion_state_dict"])
            append_dict = {k: torch.Tensor([0.0]) for k, _ in criterion.named_parameters() if k not in c_state}
            c_state.update(append_dict)
            criterion.load_state_dict(c_state)
    return start_epoch
