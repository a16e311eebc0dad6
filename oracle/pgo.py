"""TEST INFRASTRUCTURE -- CPU restatement (numpy, fp64) of the reference's pose-graph optimisation.

Follows /root/reference/common/pose_utils.py:
  skew :373-380, dpq_q :382-393, dpsq_q :395-406, dpsq_p :408-419, dqstq_t :435-443, m_rot :445-456,
  PoseGraph :458-611 (jacobian :469-509, residuals :511-553, update_on_manifold :555-577, optimize :579-611),
  PoseGraphFC :613-771 (same, over all pairs i < j in lexicographic order), optimize_poses :773-804.

The reference calls `transforms3d.quaternions` (qmult, qinverse, rotate_vector; dependency absent from
/root/reference and not installed, version unpinned in environment.yml:19).  `txq_*` below restate its
published definitions (quaternions are [w, x, y, z]):
  qmult(q1, q2)      Hamilton product
  qconjugate(q)      q * [1, -1, -1, -1]
  qinverse(q)        qconjugate(q) / dot(q, q)
  rotate_vector(v,q) (q (0, v) q*)[1:]        -- with q* the CONJUGATE, so a non-unit q scales the result by |q|^2
Pinned by tests/test_oracle_pinned_to_reference.py: the reference's own PoseGraph / PoseGraphFC classes are executed
from /root/reference with these three functions bound as `txq` and must agree with `optimize_window` below; golden
vectors (tests/golden/pgo.npz) come from that execution, including the reference's own fixture `pgo_test_poses1`
(:1146-1169).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import math

import numpy as np
import scipy.linalg as slin


# ---- transforms3d.quaternions, restated ---------------------------------------------------------
def txq_qmult(q1, q2):
    w1, x1, y1, z1 = q1
    w2, x2, y2, z2 = q2
    w = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2
    x = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2
    y = w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2
    z = w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2
    return np.array([w, x, y, z])


def txq_qconjugate(q):
    return np.array(q) * np.array([1.0, -1, -1, -1])


def txq_qinverse(q):
    return txq_qconjugate(q) / np.dot(q, q)


def txq_rotate_vector(v, q):
    varr = np.zeros((4,))
    varr[1:] = v
    return txq_qmult(q, txq_qmult(varr, txq_qconjugate(q)))[1:]


class txq:  # the namespace the reference source sees as `txq`
    qmult = staticmethod(txq_qmult)
    qconjugate = staticmethod(txq_qconjugate)
    qinverse = staticmethod(txq_qinverse)
    rotate_vector = staticmethod(txq_rotate_vector)


# ---- Jacobian blocks (pose_utils.py:373-456) ----------------------------------------------------
def skew(x):
    return np.asarray([[0, -x[2], x[1]], [x[2], 0, -x[0]], [-x[1], x[0], 0]], dtype=np.float64)


def dpq_q(p):
    """d(p q)/dq, 4x4 (:382-393)"""
    J = np.zeros((4, 4))
    J[0, 0] = p[0]
    J[0, 1:] = -p[1:]
    J[1:, 0] = p[1:]
    J[1:, 1:] = p[0] * np.eye(3) + skew(p[1:])
    return J


def dpsq_q(p):
    """d(p* q)/dq (:395-406)"""
    J = np.zeros((4, 4))
    J[0, 0] = p[0]
    J[0, 1:] = -p[1:]
    J[1:, 0] = -p[1:]
    J[1:, 1:] = p[0] * np.eye(3) - skew(p[1:])
    return J


def dpsq_p(q):
    """d(p* q)/dp (:408-419)"""
    J = np.zeros((4, 4))
    J[0, 0] = q[0]
    J[0, 1:] = q[1:]
    J[1:, 0] = q[1:]
    J[1:, 1:] = -q[0] * np.eye(3) + skew(q[1:])
    return J


def dqstq_t(q):
    """d(q* t q)/dt, 3x3 (:435-443)"""
    qv = np.asarray(q[1:]).reshape(3, 1)
    return (q[0] * q[0] - float((qv.T @ qv)[0, 0])) * np.eye(3) + 2 * (qv @ qv.T) - 2 * q[0] * skew(q[1:])


def m_rot(x):
    """Jacobian of the full quaternion w.r.t. the manifold increment at 0: columns 1..3 of dpq_q(x), 4x3 (:445-456)"""
    return dpq_q(x)[:, 1:]


def pairs(N, fc):
    """constraint pairs in the reference's order: consecutive (:487, :535) or all i<j (:642-643, :702-703)"""
    if fc:
        return [(i, j) for i in range(N) for j in range(i + 1, N)]
    return [(i, i + 1) for i in range(N - 1)]


def jacobian(z, N, fc, w_ax, w_aq, w_rx, w_rq):
    """z: [N,7].  The information factors are cholesky(I/s).T = I/sqrt(s): scalars (:596-599)."""
    pr = pairs(N, fc)
    J = np.zeros((7 * N + 7 * len(pr), 6 * N))
    row = 0
    for i in range(N):
        J[row:row + 3, 6 * i:6 * i + 3] = w_ax * np.eye(3)
        row += 3
        J[row:row + 4, 6 * i + 3:6 * i + 6] = w_aq * m_rot(z[i, 3:])
        row += 4
    for (i, j) in pr:
        dt = dqstq_t(z[i, 3:])  # the rotation dependence of the translation residual is ignored (:492-495 commented out)
        J[row:row + 3, 6 * i:6 * i + 3] = -w_rx * dt
        J[row:row + 3, 6 * j:6 * j + 3] = w_rx * dt
        row += 3
        J[row:row + 4, 6 * i + 3:6 * i + 6] = w_rq * (dpsq_p(z[j, 3:]) @ m_rot(z[i, 3:]))
        J[row:row + 4, 6 * j + 3:6 * j + 6] = w_rq * (dpsq_q(z[i, 3:]) @ m_rot(z[j, 3:]))
        row += 4
    return J


def residuals(z, poses, vos, N, fc, w_ax, w_aq, w_rx, w_rq):
    pr = pairs(N, fc)
    r = np.zeros(7 * N + 7 * len(pr))
    row = 0
    for i in range(N):
        d = z[i] - poses[i]
        r[row:row + 3] = w_ax * d[:3]
        r[row + 3:row + 7] = w_aq * d[3:]
        row += 7
    for k, (i, j) in enumerate(pr):
        v = z[j, :3] - z[i, :3]
        rt = txq_rotate_vector(v, txq_qinverse(z[i, 3:])) - vos[k, :3]
        r[row:row + 3] = w_rx * rt
        row += 3
        rq = txq_qmult(txq_qinverse(z[i, 3:]), z[j, 3:]) - vos[k, 3:]
        r[row:row + 4] = w_rq * rq
        row += 4
    return r


def update_on_manifold(z, x):
    """:555-577 -- translation += t; q <- q (x) [cos n, sinc(n/pi) qm]   (no renormalisation)"""
    for i in range(len(z)):
        z[i, :3] += x[6 * i:6 * i + 3]
        qm = x[6 * i + 3:6 * i + 6]
        n = np.linalg.norm(qm)
        dq = np.zeros(4)
        dq[0] = math.cos(n)
        dq[1:] = np.sinc(n / np.pi) * qm
        z[i, 3:] = txq_qmult(z[i, 3:], dq)


def optimize_window(poses, vos, fc=False, sax=1, saq=1, srx=1, srq=1, n_iters=10):
    """PoseGraph.optimize / PoseGraphFC.optimize (:579-611, :739-771): Gauss-Newton from z = poses;
    H = J'J, b = J'r, R = chol(H), the reference's two triangular solves (see below), update on the manifold.  poses [N,7], vos [P,7] -> [N,7]"""
    poses = np.asarray(poses, dtype=np.float64)
    vos = np.asarray(vos, dtype=np.float64)
    N = len(poses)
    w_ax, w_aq, w_rx, w_rq = (math.sqrt(1.0 / s) for s in (sax, saq, srx, srq))
    z = poses.copy()
    for _ in range(n_iters):
        J = jacobian(z, N, fc, w_ax, w_aq, w_rx, w_rq)
        r = residuals(z, poses, vos, N, fc, w_ax, w_aq, w_rx, w_rq)
        H = J.T @ J
        b = J.T @ r
        R = slin.cholesky(H)
        # :605-607 as written: `solve_triangular(R.T, -b)` with scipy's default lower=False reads only the UPPER
        # triangle of R.T, i.e. its diagonal: y = -b / diag(R), not the forward substitution.  The reference's
        # step is therefore x = R^-1 D^-1 (-b) (D = diag R) -- restated as is; it is what its results come from.
        y = slin.solve_triangular(R.T, -b)
        x = slin.solve_triangular(R, y)
        update_on_manifold(z, x)
    return z


def vos_from_targets(target_poses):
    """optimize_poses :790-797: translation difference (NOT rotated) and q0^-1 q1 of consecutive target poses"""
    t = np.asarray(target_poses, dtype=np.float64)
    vos = np.zeros((len(t) - 1, 7))
    for i in range(len(vos)):
        vos[i, :3] = t[i + 1, :3] - t[i, :3]
        vos[i, 3:] = txq_qmult(txq_qinverse(t[i, 3:]), t[i + 1, 3:])
    return vos


def optimize_poses(pred_poses, vos=None, fc_vos=False, target_poses=None, sax=1, saq=1, srx=1, srq=1):
    """:773-804"""
    if vos is None:
        if target_poses is None:
            return None  # the reference prints 'Specify either VO or target poses' and returns None
        vos = vos_from_targets(target_poses)
    return optimize_window(pred_poses, vos, fc=fc_vos, sax=sax, saq=saq, srx=srx, srq=srq)
