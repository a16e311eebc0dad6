"""Evaluation flow and error metric of the reference (host side, fp64 numpy, as the reference computes it).

Mirrors /root/reference/scripts/eval.py:153-205 (inference loop, "take the middle prediction",
un-normalisation, error statistics) and the numpy helpers it uses from
/root/reference/common/pose_utils.py: `qexp` (:319-327), `quaternion_angular_error` (:361-371);
`t_criterion` is the L2 norm of eval.py:80.  The network forward inside the loop is the HIP path
(`step_feedfwd(train=False)`); pose-graph optimisation (eval.py:177-182, `--pose_graph`) runs on the HIP path
too, batched over all windows of the loop in one launch (geomapnet_amd/pgo.py).
"""
import numpy as np

from .train import step_feedfwd


def qexp(q):
    """exponential map (3,) -> (4,): [cos n, sinc(n/pi) q]   (pose_utils.py:319-327)"""
    q = np.asarray(q)
    n = np.linalg.norm(q)
    return np.hstack((np.cos(n), np.sinc(n / np.pi) * q))


def quaternion_angular_error(q1, q2):
    """angular error between two quaternions in degrees   (pose_utils.py:361-371)"""
    d = abs(np.dot(q1, q2))
    d = min(1.0, max(-1.0, d))
    return 2 * np.arccos(d) * 180 / np.pi


def log_quaternion_angular_error(q1, q2):
    """pose_utils.py:358-359"""
    return quaternion_angular_error(qexp(q1), qexp(q2))


def t_criterion(t_pred, t_gt):
    """eval.py:80"""
    return np.linalg.norm(np.asarray(t_pred) - np.asarray(t_gt))


q_criterion = quaternion_angular_error


def to_pose7(logq_poses, pose_m, pose_s):
    """[M,6] (translation, log-quaternion) -> [M,7] (un-normalised translation, unit quaternion);
    eval.py:166-175 and :184-186"""
    p = np.asarray(logq_poses, dtype=np.float64).reshape(-1, 6)
    q = [qexp(r[3:]) for r in p]
    out = np.hstack((p[:, :3], np.asarray(q).reshape(-1, 4)))
    out[:, :3] = (out[:, :3] * np.asarray(pose_s, dtype=np.float64)) + np.asarray(pose_m, dtype=np.float64)
    return out


def pose_errors(pred_poses, targ_poses):
    """per-frame translation error (same unit as the poses) and rotation error (degrees) of [L,7] arrays
    (eval.py:192-196)"""
    t_loss = np.asarray([t_criterion(p, t) for p, t in zip(pred_poses[:, :3], targ_poses[:, :3])])
    q_loss = np.asarray([q_criterion(p, t) for p, t in zip(pred_poses[:, 3:], targ_poses[:, 3:])])
    return t_loss, q_loss


def summarize(t_loss, q_loss):
    """the numbers eval.py:202-205 prints"""
    return {"median_t": float(np.median(t_loss)), "mean_t": float(np.mean(t_loss)),
            "median_q": float(np.median(q_loss)), "mean_q": float(np.mean(q_loss))}


def evaluate(model, batches, pose_m=(0.0, 0.0, 0.0), pose_s=(1.0, 1.0, 1.0), cuda=True, pose_graph=False, fc_vos=False,
             sax=1, saq=1, srx=1, srq=1, indices_of=None, length=None):
    """Inference loop of eval.py:153-190.

    With `pose_graph` (eval.py:177-182) every target carries the window's VOs after its T absolute poses
    (`MF(include_vos=True)`: target [1,T+P,6]); predictions and VOs of all windows are collected and optimised by
    one `optimize_windows` launch, then un-normalised (eval.py:184-186 runs after the optimisation).

    `batches`: iterable of (data, target) as the reference's DataLoader yields them with batch_size 1:
    data [1,3,H,W] (PoseNet) or [1,T,3,H,W] (MapNet, `--model mapnet*`), target [1,6] / [1,T,6].  For
    every batch the MIDDLE prediction of the window is kept (eval.py:187-190: `output[len(output)/2]`).
    `indices_of(batch_idx)` (the MF dataset's `get_indices`, eval.py:158-163) with `length` = len(dataset): the middle
    prediction of batch i is WRITTEN at row `indices_of(i)[middle]` of zero-initialised [length, 7] arrays as the
    reference does -- windows clamped at the ends of a sequence repeat a middle index and overwrite each other, rows no
    window centres on stay zero -- instead of being appended one row per batch.
    Returns (summary dict, pred_poses [L,7], targ_poses [L,7])."""
    pred, targ, win_out, win_vos = [], [], [], []
    was_training = model.training
    model.eval()
    try:
        for data, target in batches:
            _, output = step_feedfwd(data, model, cuda, train=False)
            s = output.size()
            out = output.detach().cpu().numpy().reshape((-1, s[-1]))
            tgt = np.asarray(target.detach().cpu().numpy() if hasattr(target, "detach") else target).reshape((-1, s[-1]))
            if pose_graph:
                win_out.append(to_pose7(out, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)))  # qexp only
                win_vos.append(to_pose7(tgt[len(out):], (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)))
                tgt = tgt[:len(out)]
            else:
                out = to_pose7(out, pose_m, pose_s)
                pred.append(out[len(out) // 2])
            tgt = to_pose7(tgt, pose_m, pose_s)
            targ.append(tgt[len(tgt) // 2])
    finally:
        model.train(was_training)
    if pose_graph and win_out:
        from .pgo import optimize_windows
        from .posenet import engine_of
        eng = engine_of(model)  # the optimisation runs on the model's device, through the model's kernel library
        opt = optimize_windows(np.stack(win_out), np.stack(win_vos), fc_vos=fc_vos, sax=sax, saq=saq, srx=srx, srq=srq,
                               device=eng.device, binding=eng.lib)
        opt[:, :, :3] = (opt[:, :, :3] * np.asarray(pose_s, dtype=np.float64)) + np.asarray(pose_m, dtype=np.float64)
        pred = [o[len(o) // 2] for o in opt]
    pred_poses, targ_poses = np.asarray(pred), np.asarray(targ)
    if indices_of is not None:
        n = int(length) if length is not None else len(pred_poses)
        pp, tp = np.zeros((n, 7)), np.zeros((n, 7))
        for i in range(len(pred_poses)):
            idx = list(indices_of(i))
            pp[idx[len(idx) // 2]] = pred_poses[i]
            tp[idx[len(idx) // 2]] = targ_poses[i]
        pred_poses, targ_poses = pp, tp
    t_loss, q_loss = pose_errors(pred_poses, targ_poses)
    return summarize(t_loss, q_loss), pred_poses, targ_poses
