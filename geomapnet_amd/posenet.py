"""PoseNet / MapNet module facades over the gfx950 library.

Mirror of /root/reference/models/posenet.py (PoseNet :36-73, MapNet :75-97) and of the
`torchvision.models.resnet34` call sites (scripts/train.py:76, scripts/eval.py:71): same
constructors, same `forward` shapes, same state_dict keys, `.parameters()`, `.cuda()`,
`.train()/.eval()` -- but `forward` launches the HIP kernels (no autograd graph is built; the
backward pass lives inside the fused training step, see geomapnet_amd/train.py).

Parameters are zero-copy views of one flat fp32 HBM arena owned by the Engine: conv weights are
stored OHWI in the arena and exposed as OIHW-shaped permuted views, so `state_dict()` /
`load_state_dict()` interchange with reference checkpoints.
"""
import math
import os

import torch
from torch import nn

from .engine import Engine


class _Node(nn.Module):
    """pure container that gives parameters their dotted torchvision names"""


def _view(arena, e):
    n = int(e.numel)
    if e.is_int64:
        return arena[e.offset * 8: e.offset * 8 + 8].view(torch.int64).view(())
    if e.is_buffer:
        flat = arena.view(torch.float32)[e.offset: e.offset + n] if arena.dtype == torch.uint8 else arena[e.offset: e.offset + n]
    else:
        flat = arena[e.offset: e.offset + n]
    shape = [int(s) for s in e.shape[: e.ndim]]
    if e.ohwi:
        o, i, h, w = shape
        return flat.view(o, h, w, i).permute(0, 3, 1, 2)
    return flat.view(*shape)


class _ArenaModule(nn.Module):
    """nn.Module whose parameters/buffers are views into Engine arenas"""

    def _build_tree(self, engine, prefix_filter=None):
        self._engine = engine
        self._slots = []
        for e in engine.entries:
            name = e.name.decode()
            t = _view(engine.buffers if e.is_buffer else engine.params, e)
            mod = self
            parts = name.split(".")
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, _Node())
                mod = mod._modules[p]
            if e.is_buffer:
                mod.register_buffer(parts[-1], t)
            else:
                mod.register_parameter(parts[-1], nn.Parameter(t, requires_grad=False))
            self._slots.append((mod, parts[-1], e))

    def _rebind(self):
        eng = self._engine
        for mod, leaf, e in self._slots:
            t = _view(eng.buffers if e.is_buffer else eng.params, e)
            if e.is_buffer:
                mod._buffers[leaf] = t
            else:
                mod._parameters[leaf].data = t

    def _apply(self, fn, recurse=True):
        self._engine.move(fn)
        self._rebind()
        return self

    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(state_dict, strict)
        self._engine.params_touched()
        return r

    def state_dict(self, *args, **kwargs):
        """as nn.Module.state_dict; the int64 `num_batches_tracked` scalars are returned as copies because they
        share the buffer arena with the fp32 running statistics and torch.save refuses one storage viewed as
        two dtypes"""
        sd = super().state_dict(*args, **kwargs)
        for k, v in list(sd.items()):
            if torch.is_tensor(v) and v.dtype == torch.int64:
                sd[k] = v.clone()
        return sd


# ---- torchvision-style initialisation ----------------------------------------------------------
def _kaiming_normal_(t, mode):
    """nn.init.kaiming_normal_ (a=0 -> gain sqrt(2)) on an OIHW- or [out,in]-shaped view"""
    recept = 1
    for s in t.shape[2:]:
        recept *= s
    fan = t.shape[1] * recept if mode == "fan_in" else t.shape[0] * recept
    std = math.sqrt(2.0) / math.sqrt(fan)
    with torch.no_grad():
        t.copy_(torch.randn(t.shape) * std)


class ResNet34(_ArenaModule):
    """Stand-in for torchvision.models.resnet34(): carries the trunk weights (torchvision's own
    initialisation, or a model-zoo state_dict when available) that PoseNet adopts."""

    def __init__(self, pretrained=False, _binding=None):
        super().__init__()
        self._binding = _binding
        eng = Engine(2048, binding=_binding)
        self._build_tree(eng)
        # keep only the trunk under our own namespace (strip 'feature_extractor.')
        fe = self._modules.pop("feature_extractor")
        for k in list(self._modules):
            self._modules.pop(k)
        for k, m in fe._modules.items():
            self.add_module(k, m)
        self.fc.in_features = 512
        self.reset_parameters()
        if pretrained:
            self.load_zoo_weights()

    def reset_parameters(self):
        for name, p in self.named_parameters():
            if p.dim() == 4:
                _kaiming_normal_(p.data, "fan_out")
            elif ".bn" in name or "downsample.1" in name or name.startswith("bn"):
                with torch.no_grad():
                    p.data.fill_(1.0 if name.endswith("weight") else 0.0)
        for name, b in self.named_buffers():
            with torch.no_grad():
                if name.endswith("running_var"):
                    b.fill_(1.0)
                else:
                    b.zero_()
        # nn.Linear default init for the (to be replaced) fc
        with torch.no_grad():
            bound = 1.0 / math.sqrt(512)
            self.fc.weight.data.uniform_(-bound, bound)
            self.fc.bias.data.uniform_(-bound, bound)

    def load_zoo_weights(self):
        zoo = os.environ.get("TORCH_MODEL_ZOO", os.path.join("..", "data", "models"))
        path = os.path.join(zoo, "resnet34-333f7ec4.pth")
        if not os.path.isfile(path):
            raise RuntimeError("resnet34(pretrained=True): %s not found and there is no network; "
                               "place the torchvision checkpoint there or pass pretrained=False" % path)
        sd = torch.load(path, map_location="cpu")
        own = self.state_dict()
        for k, v in sd.items():
            if k in own and own[k].shape == v.shape:
                own[k].copy_(v)

    def forward(self, x):
        raise RuntimeError("ResNet34 is a weight carrier for PoseNet; call PoseNet/MapNet.forward")


def resnet34(pretrained=False, _binding=None):
    return ResNet34(pretrained=pretrained, _binding=_binding)


class PoseNet(_ArenaModule):
    def __init__(self, feature_extractor, droprate=0.5, pretrained=True, feat_dim=2048, filter_nans=False,
                 dropout_active=False, dropout_seed=0, _binding=None):
        """droprate / dropout_active: the reference calls `F.dropout(x, p=self.droprate)` without `training=`
        (models/posenet.py:68-69).  Under its pinned PyTorch 0.4.1 that default is training=False: an IDENTITY in train() and
        eval() alike -- which is what `dropout_active=False` (default) reproduces, with a warning when droprate > 0 because
        the shipped configs ask for 0.5 and get none.  `dropout_active=True` runs the operator on the device in train()
        (Philox mask keyed by `dropout_seed`, inverted scaling; never in eval()), i.e. nn.Dropout semantics."""
        super().__init__()
        self.droprate = droprate
        self.dropout_active = bool(dropout_active)
        if droprate > 0 and not self.dropout_active:
            import warnings
            warnings.warn("PoseNet(droprate=%g): dropout is an IDENTITY here, as under the reference's pinned PyTorch 0.4.1 "
                          "(F.dropout default training=False, models/posenet.py:68-69); pass dropout_active=True "
                          "(scripts: --dropout_active) to drop features on the device in training mode" % droprate,
                          stacklevel=2)
        if _binding is None:
            _binding = getattr(feature_extractor, "_binding", None)
        eng = Engine(feat_dim, binding=_binding, filter_nans=filter_nans)
        if self.dropout_active and droprate > 0:
            eng.set_dropout(droprate, dropout_seed)
        self._build_tree(eng)
        # adopt the trunk weights and buffers
        src = feature_extractor.state_dict()
        own = self.state_dict()
        with torch.no_grad():
            for k, v in src.items():
                kk = "feature_extractor." + k
                if kk in own and own[kk].shape == v.shape and not k.startswith("fc."):
                    own[kk].copy_(v)
        self.feature_extractor.fc.in_features = 512
        # initialise as the reference does (models/posenet.py:53-63)
        lin = [self.feature_extractor.fc, self.fc_xyz, self.fc_wpqr]
        with torch.no_grad():
            for m in lin:  # nn.Linear default init first (what construction leaves behind)
                bound = 1.0 / math.sqrt(m.weight.shape[1])
                m.weight.data.uniform_(-bound, bound)
                m.bias.data.uniform_(-bound, bound)
        if pretrained:
            targets = [(m.weight, m.bias) for m in lin]
        else:
            targets = [(p, None) for p in self.parameters() if p.dim() == 4]
            targets += [(m.weight, m.bias) for m in lin]
        for w, b in targets:
            _kaiming_normal_(w.data, "fan_in")
            if b is not None:
                with torch.no_grad():
                    b.data.zero_()
        eng.params_touched()

    def set_input_u8(self, mean=None, std=None):
        """Device-side ToTensor + Normalize: inputs become uint8 [N,H,W,3] (PoseNet) / [N,T,H,W,3] (MapNet) and
        (x/255 - mean)/std is applied by the conversion kernel.  `mean=None` restores fp32 [N,3,H,W] input."""
        self._engine.set_input_u8(mean, std)

    def forward(self, x):
        u8 = self._engine.input_u8 is not None
        if x.dim() != 4 or (x.shape[-1] if u8 else x.shape[1]) != 3:
            raise ValueError("PoseNet.forward expects [N,3,H,W] (or uint8 [N,H,W,3] after set_input_u8)")
        x = x.detach()
        if x.device != self._engine.device:
            raise RuntimeError("input on %s but model on %s" % (x.device, self._engine.device))
        x = x.contiguous() if u8 else x.float().contiguous()
        # dropout: the engine applies it in training mode when PoseNet(dropout_active=True); otherwise an identity, as under
        # the reference's pinned PyTorch 0.4.1 (F.dropout default training=False at models/posenet.py:68-69)
        return self._engine.forward(x, self.training)


class MapNet(nn.Module):
    def __init__(self, mapnet):
        super().__init__()
        self.mapnet = mapnet

    def forward(self, x):
        s = x.size()
        poses = self.mapnet(x.reshape(-1, *s[2:]))
        return poses.view(s[0], s[1], -1)

    def set_input_u8(self, mean=None, std=None):
        self.mapnet.set_input_u8(mean, std)

    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(state_dict, strict)
        self.mapnet._engine.params_touched()
        return r


def engine_of(model):
    m = model.mapnet if isinstance(model, MapNet) else model
    if not isinstance(m, PoseNet):
        raise TypeError("expected geomapnet_amd PoseNet/MapNet, got %r" % type(model))
    return m._engine
