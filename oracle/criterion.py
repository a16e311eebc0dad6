"""Oracle: pose criteria (TEST INFRASTRUCTURE).

Restates /root/reference/common/criterion.py:
  PoseNetCriterion       :33-52
  MapNetCriterion        :54-109   (uses calc_vos_simple, pose_utils.py:234-246)
  MapNetOnlineCriterion  :111-184  (uses calc_vos, pose_utils.py:248-260; gps_mode :166,173-180)
All share one form: exp(-s) * L1mean(pred_part, targ_part) + s per term, with learnable scalar
log-weights sax/saq (absolute) and srx/srq (relative) created as nn.Parameter([1]) whose
requires_grad is the learn flag (:39-40, :71-74, :131-134).  `T = s[1] / 2` at :150 is
Python-2 integer division and is restated as `//`.
"""
import torch
from torch import nn

from . import pose_math


def _weighted_l1(s, a, b):
    return torch.exp(-s) * torch.mean(torch.abs(a - b)) + s


class _PoseCriterion(nn.Module):
    def __init__(self, sax, saq, srx=None, srq=None, learn_beta=False, learn_gamma=False):
        super().__init__()
        self.sax = nn.Parameter(torch.Tensor([sax]), requires_grad=learn_beta)
        self.saq = nn.Parameter(torch.Tensor([saq]), requires_grad=learn_beta)
        if srx is not None:
            self.srx = nn.Parameter(torch.Tensor([srx]), requires_grad=learn_gamma)
            self.srq = nn.Parameter(torch.Tensor([srq]), requires_grad=learn_gamma)

    def _abs_term(self, pred, targ):
        p, g = pred.reshape(-1, pred.shape[-1]), targ.reshape(-1, targ.shape[-1])
        return _weighted_l1(self.sax, p[:, :3], g[:, :3]) + _weighted_l1(self.saq, p[:, 3:], g[:, 3:])


class PoseNetCriterion(_PoseCriterion):
    def __init__(self, sax=0.0, saq=0.0, learn_beta=False):
        super().__init__(sax, saq, learn_beta=learn_beta)

    def forward(self, pred, targ):
        return self._abs_term(pred, targ)


class MapNetCriterion(_PoseCriterion):
    def __init__(self, sax=0.0, saq=0.0, srx=0.0, srq=0.0, learn_beta=False, learn_gamma=False):
        super().__init__(sax, saq, srx, srq, learn_beta, learn_gamma)

    def forward(self, pred, targ):
        pv = pose_math.calc_vos_simple(pred).reshape(-1, pred.shape[-1])
        gv = pose_math.calc_vos_simple(targ).reshape(-1, targ.shape[-1])
        rel = _weighted_l1(self.srx, pv[:, :3], gv[:, :3]) + _weighted_l1(self.srq, pv[:, 3:], gv[:, 3:])
        return self._abs_term(pred, targ) + rel


class MapNetOnlineCriterion(_PoseCriterion):
    def __init__(self, sax=0.0, saq=0.0, srx=0.0, srq=0.0, learn_beta=False, learn_gamma=False, gps_mode=False):
        super().__init__(sax, saq, srx, srq, learn_beta, learn_gamma)
        self.gps_mode = gps_mode

    def forward(self, pred, targ):
        T = pred.shape[1] // 2
        loss = self._abs_term(pred[:, :T].contiguous(), targ[:, :T].contiguous())
        pv, gv = pred[:, T:].contiguous(), targ[:, T:].contiguous()
        if self.gps_mode:
            pv, gv = pv.reshape(-1, pv.shape[-1]), gv.reshape(-1, gv.shape[-1])
            return loss + _weighted_l1(self.srx, pv[:, :2], gv[:, :2])
        pv = pose_math.calc_vos(pv).reshape(-1, pv.shape[-1])
        gv = gv.reshape(-1, gv.shape[-1])
        return loss + _weighted_l1(self.srx, pv[:, :3], gv[:, :3]) + _weighted_l1(self.srq, pv[:, 3:], gv[:, 3:])
