"""Pose-graph optimisation at inference -- the reference's `optimize_poses` / `PoseGraph` / `PoseGraphFC`
(/root/reference/common/pose_utils.py:458-804, called per window at scripts/eval.py:177-182), on the HIP path.

Same names, arguments and results as the reference; the arithmetic runs in `mn_pgo_optimize` (csrc/pgo.h): one
wavefront per window, fp64, all windows of a call in a single launch (`optimize_windows` is the batched form the
evaluation loop uses; the reference's one-window signatures are thin wrappers over it).  No CPU fallback: without
the HIP library / a GPU this raises.
"""
import ctypes as C

import numpy as np
import torch

from . import _binding

MAX_POSES = 12


def _lib(binding):
    return binding if binding is not None else _binding.hip()


def optimize_windows(pred_poses, vos, fc_vos=False, sax=1, saq=1, srx=1, srq=1, n_iters=10, device="cuda", binding=None):
    """pred_poses [W,N,7], vos [W,P,7] (P = N-1, or N(N-1)/2 with fc_vos) -> optimised poses [W,N,7] (numpy fp64).
    Raises numpy.linalg.LinAlgError if any window's normal matrix is not positive definite (as scipy's cholesky
    does inside the reference, :605)."""
    pred = np.ascontiguousarray(pred_poses, dtype=np.float64)
    v = np.ascontiguousarray(vos, dtype=np.float64)
    if pred.ndim != 3 or pred.shape[2] != 7:
        raise ValueError("pred_poses must be [W,N,7]")
    W, N = pred.shape[0], pred.shape[1]
    P = N * (N - 1) // 2 if fc_vos else N - 1
    if v.ndim != 3 or v.shape[0] != W or v.shape[2] != 7 or v.shape[1] < P:
        raise ValueError("vos must be [W,%d,7] for %d poses per window (fc_vos=%s)" % (P, N, bool(fc_vos)))
    v = np.ascontiguousarray(v[:, :P])  # the chain graph reads only its first N-1 rows (as the reference does)
    lib = _lib(binding)
    dev = torch.device(device)
    d_pred = torch.from_numpy(pred).to(dev)
    d_vos = torch.from_numpy(v).to(dev)
    d_out = torch.empty_like(d_pred)
    d_status = torch.empty(W, dtype=torch.int32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream) if dev.type == "cuda" else None
    lib.check(lib.pgo_optimize(_binding.ptr(d_pred), _binding.ptr(d_vos), _binding.ptr(d_out), _binding.ptr(d_status), W, N,
                               1 if fc_vos else 0, float(sax), float(saq), float(srx), float(srq), int(n_iters), stream))
    status = d_status.cpu().numpy()
    if status.any():
        raise np.linalg.LinAlgError("pose graph: normal matrix not positive definite in window(s) %s"
                                    % np.flatnonzero(status)[:8].tolist())
    return d_out.cpu().numpy()


class PoseGraph:
    """pose_utils.py:458-611 (consecutive-frame constraints)"""
    fc = False

    def __init__(self, device="cuda", binding=None):
        self.N = 0
        self.z = np.zeros((0, 0))
        self._device, self._binding = device, binding

    def optimize(self, poses, vos, sax=1, saq=1, srx=1, srq=1, n_iters=10):
        poses = np.asarray(poses, dtype=np.float64)
        self.N = len(poses)
        out = optimize_windows(poses[None], np.asarray(vos, dtype=np.float64)[None], fc_vos=self.fc, sax=sax, saq=saq,
                               srx=srx, srq=srq, n_iters=n_iters, device=self._device, binding=self._binding)[0]
        self.z = out.reshape((-1, 1)).copy()
        return out


class PoseGraphFC(PoseGraph):
    """pose_utils.py:613-771 (all pairs i < j)"""
    fc = True


def vos_from_target_poses(target_poses):
    """optimize_poses :790-797: VOs from target poses -- translation difference (not rotated) and q0^-1 q1 with
    transforms3d's qinverse = conjugate / |q|^2"""
    t = np.asarray(target_poses, dtype=np.float64)
    vos = np.zeros((len(t) - 1, 7))
    vos[:, :3] = t[1:, :3] - t[:-1, :3]
    q0, q1 = t[:-1, 3:], t[1:, 3:]
    qi = q0 * np.array([1.0, -1, -1, -1]) / np.sum(q0 * q0, axis=1, keepdims=True)
    w1, x1, y1, z1 = qi.T
    w2, x2, y2, z2 = q1.T
    vos[:, 3] = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2
    vos[:, 4] = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2
    vos[:, 5] = w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2
    vos[:, 6] = w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2
    return vos


def optimize_poses(pred_poses, vos=None, fc_vos=False, target_poses=None, sax=1, saq=1, srx=1, srq=1, device="cuda",
                   binding=None):
    """pose_utils.py:773-804"""
    pgo = (PoseGraphFC if fc_vos else PoseGraph)(device=device, binding=binding)
    if vos is None:
        if target_poses is not None:
            vos = vos_from_target_poses(target_poses)
        else:
            print("Specify either VO or target poses")
            return None
    return pgo.optimize(poses=pred_poses, vos=vos, sax=sax, saq=saq, srx=srx, srq=srq)
