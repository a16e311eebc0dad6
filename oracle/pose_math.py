"""Oracle: log-quaternion pose algebra (TEST INFRASTRUCTURE).

Batched restatement of the torch section of /root/reference/common/pose_utils.py (:21-304).
The reference evaluates these per consecutive pose pair inside Python loops with N=1 tensors
(calc_vos :248-260, calc_vos_simple :234-246); its `qmult` is only well-formed for N=1
(vdot returns shape (N,), :28-30 vs :51-54).  Here every function carries a leading batch of
independent pairs and applies the N=1 semantics row-wise with the same primitive ops
(norm, clamp, sin/cos/acos, cross, division), so autograd reproduces the reference gradients
including its clamp sub-gradients and 0*inf -> NaN cases.

Also: numpy `qexp` (:319-327) and `quaternion_angular_error` (:361-371) used by the
evaluation metric (scripts/eval.py:172-205).
"""
import numpy as np
import torch


def _l2(v):
    return torch.norm(v, p=2, dim=-1, keepdim=True)


def qexp_t(w):
    """log-quaternion [...,3] -> unit quaternion [...,4]; pose_utils.py:73-84."""
    n = torch.clamp(_l2(w), min=1e-8)
    return torch.cat((torch.cos(n), w * torch.sin(n) / n), dim=-1)


def qlog_t(q):
    """quaternion [...,4] -> log-quaternion [...,3]; pose_utils.py:86-96."""
    v = q[..., 1:]
    n = torch.clamp(_l2(v), min=1e-8)
    return v * torch.acos(torch.clamp(q[..., :1], min=-1.0, max=1.0)) / n


def qinv(q):
    """conjugate; pose_utils.py:64-71."""
    return torch.cat((q[..., :1], -q[..., 1:]), dim=-1)


def qmult(a, b):
    """Hamilton product followed by L2 normalisation (pose_utils.py:44-62, normalize :32-42)."""
    a0, av = a[..., :1], a[..., 1:]
    b0, bv = b[..., :1], b[..., 1:]
    s = a0 * b0 - (av * bv).sum(dim=-1, keepdim=True)
    v = av * b0 + bv * a0 + torch.cross(av, bv, dim=-1)
    q = torch.cat((s, v), dim=-1)
    return q / torch.norm(q, p=2, dim=-1).unsqueeze(-1)


def rotate_vec_by_q(t, q):
    """t + 2 q0 (qv x t) + 2 qv x (qv x t); pose_utils.py:120-132."""
    q0, qv = q[..., :1], q[..., 1:]
    b = torch.cross(qv, t, dim=-1)
    c = 2 * torch.cross(qv, b, dim=-1)
    return t + 2 * b * q0 + c


def compose_pose_quaternion(p1, p2):
    """pose_utils.py:134-146 (poses are [t(3), q(4)])."""
    q = qmult(p1[..., 3:], p2[..., 3:])
    t = p1[..., :3] + rotate_vec_by_q(p2[..., :3], p1[..., 3:])
    return torch.cat((t, q), dim=-1)


def invert_pose_quaternion(p):
    """pose_utils.py:148-157."""
    qi = qinv(p[..., 3:])
    return torch.cat((-rotate_vec_by_q(p[..., :3], qi), qi), dim=-1)


def calc_vo(p0, p1):
    """pose_utils.py:159-165."""
    return compose_pose_quaternion(invert_pose_quaternion(p0), p1)


def calc_vo_logq(p0, p1):
    """relative pose of p1 in the frame of p0, log-quaternion in/out; pose_utils.py:167-179."""
    a = torch.cat((p0[..., :3], qexp_t(p0[..., 3:])), dim=-1)
    b = torch.cat((p1[..., :3], qexp_t(p1[..., 3:])), dim=-1)
    vo = calc_vo(a, b)
    return torch.cat((vo[..., :3], qlog_t(vo[..., 3:])), dim=-1)


def calc_vos_simple(poses):
    """[N,T,6] -> [N,T-1,6] plain difference of consecutive 6-vectors; pose_utils.py:234-246."""
    return poses[:, 1:] - poses[:, :-1]


def calc_vos(poses):
    """[N,T,6] -> [N,T-1,6] via calc_vo_logq on consecutive pairs; pose_utils.py:248-260."""
    return calc_vo_logq(poses[:, :-1], poses[:, 1:])


# ---- numpy side (evaluation metric) ---------------------------------------------------------
def qexp_np(w):
    """pose_utils.py:319-327: [cos n, sinc(n/pi) * w]."""
    w = np.asarray(w, dtype=np.float64)
    n = np.linalg.norm(w)
    return np.hstack((np.cos(n), np.sinc(n / np.pi) * w))


def qlog_np(q):
    """pose_utils.py:307-317."""
    q = np.asarray(q, dtype=np.float64)
    if np.all(q[1:] == 0):
        return np.zeros(3)
    return np.arccos(q[0]) * q[1:] / np.linalg.norm(q[1:])


def quaternion_angular_error(q1, q2):
    """pose_utils.py:361-371, degrees."""
    d = abs(float(np.dot(q1, q2)))
    d = min(1.0, max(-1.0, d))
    return 2 * np.arccos(d) * 180 / np.pi


def calc_vos_safe_np(poses):
    """numpy restatement of calc_vos_safe (pose_utils.py:219-232,276-288): VO targets for MapNet++:
    t = R(q0*) (t1 - t0), q = log(normalise(q0* q1)).  poses: [N,T,6] float array."""
    p = torch.as_tensor(np.asarray(poses), dtype=torch.float64)
    p0, p1 = p[:, :-1], p[:, 1:]
    q0i = qinv(qexp_t(p0[..., 3:]))
    t = rotate_vec_by_q(p1[..., :3] - p0[..., :3], q0i)
    q = qmult(q0i, qexp_t(p1[..., 3:])).numpy()
    out = np.zeros((p.shape[0], p.shape[1] - 1, 6))
    out[..., :3] = t.numpy()
    for n in range(out.shape[0]):
        for i in range(out.shape[1]):
            out[n, i, 3:] = qlog_np(q[n, i])
    return out
