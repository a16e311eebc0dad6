"""Criterion facades: same constructors/attributes as /root/reference/common/criterion.py
(PoseNetCriterion :33-52, MapNetCriterion :54-109, MapNetOnlineCriterion :111-184); `forward`
launches the fused loss kernel (geomapnet_amd/csrc/criterion.h) and returns a 0-dim tensor.
The loss functions are fixed to the reference's defaults (nn.L1Loss for translation and
rotation), which is what every shipped script uses (scripts/train.py:87-99).

The four log-weights are nn.Parameter([1]) with requires_grad = learn flag, as in the reference.
When a criterion is used in a fused training step its parameters are re-pointed at the tail of
the model's parameter arena so the fused Adam updates them in place (train.step_feedfwd).
"""
import ctypes as C

import torch
from torch import nn

from . import _binding
from ._binding import ptr
from .engine import MODE_GPS, MODE_MAPNET, MODE_ONLINE, MODE_POSENET, _stream

_NAMES = ("sax", "saq", "srx", "srq")


class _Criterion(nn.Module):
    mode = None

    def __init__(self, t_loss_fn=None, q_loss_fn=None, sax=0.0, saq=0.0, srx=0.0, srq=0.0, learn_beta=False,
                 learn_gamma=False, has_rel=True, _binding=None):
        super().__init__()
        for fn in (t_loss_fn, q_loss_fn):
            if fn is not None and not isinstance(fn, nn.L1Loss):
                raise NotImplementedError("the fused HIP criterion implements the reference default nn.L1Loss only")
        self._lib = _binding
        self._store = torch.tensor([sax, saq, srx, srq], dtype=torch.float32)
        self.sax = nn.Parameter(self._store[0:1], requires_grad=learn_beta)
        self.saq = nn.Parameter(self._store[1:2], requires_grad=learn_beta)
        if has_rel:
            self.srx = nn.Parameter(self._store[2:3], requires_grad=learn_gamma)
            self.srq = nn.Parameter(self._store[3:4], requires_grad=learn_gamma)
        self._has_rel = has_rel

    # the four scalars always live contiguously in self._store (possibly a slice of a model arena)
    def _rebind(self, store):
        with torch.no_grad():
            store.copy_(self._store.to(store.device))
        self._store = store
        self.sax.data, self.saq.data = store[0:1], store[1:2]
        if self._has_rel:
            self.srx.data, self.srq.data = store[2:3], store[3:4]

    def _apply(self, fn, recurse=True):
        self._rebind(fn(self._store.clone()))
        return self

    @property
    def learn_beta(self):
        return self.sax.requires_grad

    @property
    def learn_gamma(self):
        return self._has_rel and self.srx.requires_grad

    def _windows_T(self, pred, targ):
        raise NotImplementedError

    def check_batch(self, n, frames, targ):
        """shape validation of a fused training step's target against the prediction [n, frames, 6] ([n, 6] for
        PoseNet) the model will produce -- the same checks `forward` applies; raises ValueError"""
        shape = (n, 6) if self.mode == MODE_POSENET else (n, frames, 6)
        return self._windows_T(torch.empty(shape, device="meta"), targ)

    def forward(self, pred, targ):
        lib = self._lib if self._lib is not None else _binding.hip()
        pred = pred.detach().float().contiguous()
        targ = targ.detach().float().contiguous().to(pred.device)
        if self._store.device != pred.device:
            self._rebind(self._store.to(pred.device))
        if lib.backend_name == "hip" and not pred.is_cuda:
            raise _binding.MapNetHipError("criterion inputs must be on the GPU; there is no CPU fallback")
        n, t = self._windows_T(pred, targ)
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        lib.check(lib.op_criterion(self.mode, n, t, ptr(pred), ptr(targ), ptr(self._store), ptr(loss), None, None, None,
                                   C.c_float(1.0), _stream(pred)))
        return loss[0]


class PoseNetCriterion(_Criterion):
    mode = MODE_POSENET

    def __init__(self, t_loss_fn=None, q_loss_fn=None, sax=0.0, saq=0.0, learn_beta=False, _binding=None):
        super().__init__(t_loss_fn, q_loss_fn, sax, saq, 0.0, 0.0, learn_beta, False, has_rel=False, _binding=_binding)

    def _windows_T(self, pred, targ):
        if pred.dim() != 2 or pred.shape[1] != 6 or targ.shape != pred.shape:
            raise ValueError("PoseNetCriterion expects pred/targ [N,6]")
        return pred.shape[0], 1


class MapNetCriterion(_Criterion):
    mode = MODE_MAPNET

    def __init__(self, t_loss_fn=None, q_loss_fn=None, sax=0.0, saq=0.0, srx=0.0, srq=0.0, learn_beta=False,
                 learn_gamma=False, _binding=None):
        super().__init__(t_loss_fn, q_loss_fn, sax, saq, srx, srq, learn_beta, learn_gamma, _binding=_binding)

    def _windows_T(self, pred, targ):
        if pred.dim() != 3 or pred.shape[2] != 6 or targ.shape != pred.shape or pred.shape[1] < 2:
            raise ValueError("MapNetCriterion expects pred/targ [N,T,6] with T >= 2")
        return pred.shape[0], pred.shape[1]


class MapNetOnlineCriterion(_Criterion):
    def __init__(self, t_loss_fn=None, q_loss_fn=None, sax=0.0, saq=0.0, srx=0.0, srq=0.0, learn_beta=False,
                 learn_gamma=False, gps_mode=False, _binding=None):
        super().__init__(t_loss_fn, q_loss_fn, sax, saq, srx, srq, learn_beta, learn_gamma, _binding=_binding)
        self.gps_mode = gps_mode

    @property
    def mode(self):
        return MODE_GPS if self.gps_mode else MODE_ONLINE

    def _windows_T(self, pred, targ):
        if pred.dim() != 3 or pred.shape[2] != 6 or pred.shape[1] % 2 or pred.shape[1] < 4:
            raise ValueError("MapNetOnlineCriterion expects pred [N,2T,6] with T >= 2")
        T = pred.shape[1] // 2  # Python-2 integer division at common/criterion.py:150
        want = 2 * T if self.gps_mode else 2 * T - 1
        if targ.shape[0] != pred.shape[0] or targ.shape[1] != want or targ.shape[2] != 6:
            raise ValueError("MapNetOnlineCriterion expects targ [N,%d,6]" % want)
        return pred.shape[0], T
