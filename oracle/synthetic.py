"""Synthetic batches for parity tests and the benchmark (TEST INFRASTRUCTURE for the parity
part; bench.py generates its device-side inputs itself with the same recipe).

Recipe from SURVEY.md section 8(d): images ~ N(0,1) fp32 (the post-`Normalize` distribution,
scripts/train.py:126-127); translations ~ N(0,1) (normalised coordinates,
common/pose_utils.py:354-355); rotations as log-quaternions = unit axis * half-angle with
half-angle ~ U(0.05, 1.2) rad (hemisphere-constrained like :347, away from the 0 and pi
singularities).  MapNet++ targets [N, 2T-1, 6]: T absolute poses then T-1 VO targets computed
with calc_vos_safe semantics (:276-288) from an independent synthetic window.
"""
import numpy as np
import torch

from . import pose_math


def _poses(gen, n, t):
    tr = torch.randn(n, t, 3, generator=gen)
    axis = torch.randn(n, t, 3, generator=gen)
    axis = axis / axis.norm(dim=-1, keepdim=True)
    half = 0.05 + 1.15 * torch.rand(n, t, 1, generator=gen)
    return torch.cat((tr, axis * half), dim=-1)


def make_batch(mode, n, h, w, t=3, seed=7, gps_mode=False):
    """mode in {'posenet','mapnet','mapnet++'} -> (images, targets) fp32 CPU tensors."""
    gen = torch.Generator().manual_seed(seed)
    if mode == "posenet":
        return torch.randn(n, 3, h, w, generator=gen), _poses(gen, n, 1)[:, 0]
    if mode == "mapnet":
        return torch.randn(n, t, 3, h, w, generator=gen), _poses(gen, n, t)
    if mode == "mapnet++":
        imgs = torch.randn(n, 2 * t, 3, h, w, generator=gen)
        absp = _poses(gen, n, t)
        rel_src = _poses(gen, n, t)
        if gps_mode:
            targ = torch.cat((absp, rel_src), dim=1)  # [N, 2T, 6], absolute translations
        else:
            vos = torch.from_numpy(pose_math.calc_vos_safe_np(rel_src.numpy())).float()
            targ = torch.cat((absp, vos), dim=1)  # [N, 2T-1, 6]
        return imgs, targ
    raise ValueError(mode)
