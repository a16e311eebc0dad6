"""Generates the golden vectors under tests/golden/ by EXECUTING THE REFERENCE's own modules
(models/posenet.py, common/criterion.py, common/pose_utils.py:1-304 from /root/reference, via
oracle/ref_loader.py).  Run in the build container only:  python tests/golden/make_golden.py
The reference ships no golden vectors for this path; these files are what pins parity on the
GPU box, where /root/reference does not exist.
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
warnings.filterwarnings("ignore")

import oracle  # noqa: E402
from oracle import ref_loader  # noqa: E402
from oracle.synthetic import _poses  # noqa: E402


def crit_case(crit, pred, targ, names):
    crit = crit.double()
    p = pred.double().clone().requires_grad_(True)
    loss = crit(p, targ.double())
    loss.backward()
    out = {"pred": pred.numpy(), "targ": targ.numpy(), "loss": np.float64(loss.item()), "dpred": p.grad.numpy()}
    for n in names:
        out["s_" + n] = np.float64(getattr(crit, n).item())
        out["d_" + n] = np.float64(getattr(crit, n).grad.item())
    return out


def make_eval_metric(ns):
    """evaluation metric of scripts/eval.py computed with the reference's numpy helpers
    (common/pose_utils.py:319-327 qexp, :361-371 quaternion_angular_error; eval.py:80 t_criterion,
    :166-175 qexp of predictions and targets, :184-186 un-normalisation, :192-205 statistics)"""
    P = ns.pose_utils_np
    rng = np.random.RandomState(20260925)
    L = 257
    targ = np.concatenate((rng.randn(L, 3), np.zeros((L, 3))), axis=1)
    axis = rng.randn(L, 3)
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    targ[:, 3:] = axis * rng.uniform(0.05, 1.2, size=(L, 1))
    pred = targ + np.concatenate((0.2 * rng.randn(L, 3), 0.05 * rng.randn(L, 3)), axis=1)
    pred[0] = targ[0]                 # identical pose: zero error
    pred[1, 3:] = 0.0                 # zero rotation vector: sinc(0) branch of qexp
    pred[2, 3:] = -targ[2, 3:]        # opposite rotation
    pose_m, pose_s = np.array([1.5, -0.3, 12.0]), np.array([3.0, 0.7, 20.0])
    pred7 = np.hstack((pred[:, :3], np.asarray([P.qexp(p[3:]) for p in pred])))
    targ7 = np.hstack((targ[:, :3], np.asarray([P.qexp(p[3:]) for p in targ])))
    pred7[:, :3] = (pred7[:, :3] * pose_s) + pose_m
    targ7[:, :3] = (targ7[:, :3] * pose_s) + pose_m
    t_loss = np.asarray([np.linalg.norm(p - t) for p, t in zip(pred7[:, :3], targ7[:, :3])])
    q_loss = np.asarray([P.quaternion_angular_error(p, t) for p, t in zip(pred7[:, 3:], targ7[:, 3:])])
    lq = np.asarray([P.log_quaternion_angular_error(p[3:], t[3:]) for p, t in zip(pred, targ)])
    np.savez_compressed(os.path.join(HERE, "eval_metric.npz"), pred=pred, targ=targ, pose_m=pose_m, pose_s=pose_s, pred7=pred7,
                        targ7=targ7, t_loss=t_loss, q_loss=q_loss, logq_err=lq,
                        stats=np.array([np.median(t_loss), np.mean(t_loss), np.median(q_loss), np.mean(q_loss)]))


def make_batch_construction(ns):
    """window indices of MF.get_indices (dataset_loaders/composite.py:60-75) and VO targets of calc_vos_safe /
    calc_vos_simple (common/pose_utils.py:219-246,276-288), executed from the reference sources"""
    import types
    out = {}
    cases = []
    for steps, skip, nodup, L in ((3, 10, False, 100), (3, 10, True, 100), (2, 1, False, 7), (5, 3, True, 40), (4, 2, False, 9),
                                  (3, 1, True, 3)):
        self = types.SimpleNamespace(variable_skip=False, skip=skip, steps=steps, no_duplicates=nodup, dset=list(range(L)))
        for index in sorted({0, 1, L // 2, L - 2, L - 1}):
            if index < 0:
                continue
            cases.append([steps, skip, int(nodup), L, index] + list(ns.mf_get_indices(self, index)) + [-1] * (8 - steps))
    out["get_indices"] = np.asarray(cases, dtype=np.int64)
    gen = torch.Generator().manual_seed(4321)
    poses = _poses(gen, 6, 4)
    out["poses"] = poses.numpy()
    out["vos_safe"] = ns.pose_utils.calc_vos_safe(poses).numpy()
    out["vos_simple"] = ns.pose_utils.calc_vos_simple(poses).numpy()
    np.savez_compressed(os.path.join(HERE, "batch_construction.npz"), **out)


def pgo_window(rng, N, fc, noise):
    """a smooth ground-truth trajectory, its exact VOs (chain or all pairs) and a noisy prediction of it"""
    from oracle import pgo as opgo
    t = np.cumsum(rng.normal(size=(N, 3)) * 0.3, axis=0)
    q = np.zeros((N, 4))
    q[0] = rng.normal(size=4)
    q[0] /= np.linalg.norm(q[0])
    for i in range(1, N):
        dq = np.r_[1.0, rng.normal(size=3) * 0.1]
        q[i] = opgo.txq_qmult(q[i - 1], dq / np.linalg.norm(dq))
    gt = np.hstack([t, q])
    pr = opgo.pairs(N, fc)
    vos = np.zeros((len(pr), 7))
    for k, (i, j) in enumerate(pr):
        vos[k, :3] = opgo.txq_rotate_vector(gt[j, :3] - gt[i, :3], opgo.txq_qinverse(gt[i, 3:]))
        vos[k, 3:] = opgo.txq_qmult(opgo.txq_qinverse(gt[i, 3:]), gt[j, 3:])
    vos[:, :3] += rng.normal(size=(len(pr), 3)) * 0.2 * noise
    pred = gt.copy()
    pred[:, :3] += rng.normal(size=(N, 3)) * noise
    for i in range(N):
        dq = np.r_[1.0, rng.normal(size=3) * noise]
        pred[i, 3:] = opgo.txq_qmult(pred[i, 3:], dq / np.linalg.norm(dq))
    return pred, vos, gt


def make_pgo(ns):
    """pose-graph optimisation (common/pose_utils.py:458-804) executed from the reference source: its own fixture
    `pgo_test_poses1` (:1146-1169, as `test_pgo` :1179-1195 runs it: fully connected VOs handed to the chain
    graph, which reads the first two) and seeded windows for the chain / fully connected graphs, several window
    lengths and covariances, and `optimize_poses` with VOs derived from target poses"""
    P = ns.pgo
    out = {}
    poses, vos = P.pgo_test_poses1()
    out["fixture/poses"], out["fixture/vos"] = poses, vos
    out["fixture/opt"] = P.PoseGraph().optimize(poses, vos)
    rng = np.random.default_rng(20180618)
    cases = []
    for tag, N, fc, sig, nwin, noise in (("chain7", 7, False, (1, 1, 1, 1), 6, 0.05), ("fc7", 7, True, (1, 1, 1, 1), 6, 0.05),
                                         ("chain7_sig", 7, False, (0.5, 2.0, 0.25, 4.0), 3, 0.1),
                                         ("fc7_sig", 7, True, (2.0, 0.5, 20.0, 20.0), 3, 0.1),
                                         ("chain2", 2, False, (1, 1, 1, 1), 2, 0.05), ("fc3", 3, True, (1, 1, 1, 1), 2, 0.05),
                                         ("chain12", 12, False, (1, 1, 1, 1), 2, 0.03), ("fc12", 12, True, (1, 1, 3.0, 3.0), 2, 0.03)):
        preds, voss, opts = [], [], []
        for _ in range(nwin):
            pred, v, _gt = pgo_window(rng, N, fc, noise)
            g = P.PoseGraphFC() if fc else P.PoseGraph()
            opts.append(g.optimize(pred, v, sax=sig[0], saq=sig[1], srx=sig[2], srq=sig[3]))
            preds.append(pred)
            voss.append(v)
        out[tag + "/pred"], out[tag + "/vos"], out[tag + "/opt"] = np.stack(preds), np.stack(voss), np.stack(opts)
        out[tag + "/cfg"] = np.asarray([N, int(fc)] + list(sig), dtype=np.float64)
        cases.append(tag)
    pred, _v, gt = pgo_window(rng, 7, False, 0.05)
    out["from_targets/pred"], out["from_targets/targ"] = pred, gt
    out["from_targets/opt"] = P.optimize_poses(pred, target_poses=gt, srx=0.5, srq=0.5)
    out["cases"] = np.asarray(cases)
    np.savez_compressed(os.path.join(HERE, "pgo.npz"), **out)


def main():
    ns = ref_loader.load()
    if sys.argv[1:] == ["pgo"]:
        make_pgo(ns)
        print("wrote pgo.npz")
        return
    if sys.argv[1:] == ["eval_metric"]:
        make_eval_metric(ns)
        print("wrote eval_metric.npz")
        return
    if sys.argv[1:] == ["batch_construction"]:
        make_batch_construction(ns)
        print("wrote batch_construction.npz")
        return
    make_eval_metric(ns)
    make_batch_construction(ns)
    make_pgo(ns)
    C = ns.criterion
    gen = torch.Generator().manual_seed(1234)
    cases = {}
    # --- criteria on prediction/target pose sets (pred = target + noise) --------------------
    for tag, n in (("n5", 5), ("n64", 64)):
        targ = _poses(gen, n, 3)
        pred = targ + 0.3 * torch.randn(n, 3, 6, generator=gen)
        cases["posenet_" + tag] = crit_case(C.PoseNetCriterion(sax=0.3, saq=-3.0, learn_beta=True), pred[:, 0], targ[:, 0],
                                            ["sax", "saq"])
        cases["mapnet_" + tag] = crit_case(
            C.MapNetCriterion(sax=0.3, saq=-3.0, srx=-0.2, srq=-3.0, learn_beta=True, learn_gamma=True), pred, targ,
            ["sax", "saq", "srx", "srq"])
        _, targ_o = oracle.make_batch("mapnet++", n, 2, 2, seed=100 + n)
        pred_o = torch.cat((targ_o[:, :3], _poses(gen, n, 3)), dim=1) + 0.2 * torch.randn(n, 6, 6, generator=gen)
        cases["online_" + tag] = crit_case(
            ns.MapNetOnlineCriterionPy3(sax=0.0, saq=-3.0, srx=0.1, srq=-3.0, learn_beta=True, learn_gamma=True), pred_o,
            targ_o, ["sax", "saq", "srx", "srq"])
        _, targ_g = oracle.make_batch("mapnet++", n, 2, 2, seed=200 + n, gps_mode=True)
        pred_g = targ_g + 0.2 * torch.randn(n, 6, 6, generator=gen)
        cases["gps_" + tag] = crit_case(
            ns.MapNetOnlineCriterionPy3(sax=0.0, saq=-3.0, srx=0.1, srq=-3.0, learn_beta=True, learn_gamma=True,
                                        gps_mode=True), pred_g, targ_g, ["sax", "saq", "srx"])
    # ragged edge: a single window
    targ = _poses(gen, 1, 3)
    pred = targ + 0.3 * torch.randn(1, 3, 6, generator=gen)
    cases["mapnet_n1"] = crit_case(C.MapNetCriterion(saq=-3.0, srq=-3.0, learn_beta=True, learn_gamma=True), pred, targ,
                                   ["sax", "saq", "srx", "srq"])
    # NaN hazard: two consecutive identical predicted rotations in the VO half
    _, targ_o = oracle.make_batch("mapnet++", 2, 2, 2, seed=5)
    pred_o = torch.cat((targ_o[:, :3], _poses(gen, 2, 3)), dim=1)
    pred_o[0, 4, 3:] = pred_o[0, 3, 3:]
    cases["online_nan"] = crit_case(
        ns.MapNetOnlineCriterionPy3(saq=-3.0, srq=-3.0, learn_beta=True, learn_gamma=True), pred_o, targ_o,
        ["sax", "saq", "srx", "srq"])
    flat = {}
    for k, d in cases.items():
        for kk, v in d.items():
            flat[k + "/" + kk] = v
    np.savez_compressed(os.path.join(HERE, "criteria.npz"), **flat)

    # --- pose algebra: calc_vos, calc_vos_simple + a VJP --------------------------------------
    P = ns.pose_utils
    poses = _poses(gen, 7, 4).double()
    R = torch.randn(7, 3, 6, generator=gen).double()
    p = poses.clone().requires_grad_(True)
    vos = P.calc_vos(p)
    (vos * R).sum().backward()
    pa = {"poses": poses.numpy(), "cot": R.numpy(), "calc_vos": vos.detach().numpy(), "calc_vos_vjp": p.grad.numpy(),
          "calc_vos_simple": P.calc_vos_simple(poses).numpy()}
    lq = poses[:, 0, 3:]
    pa["qexp"] = P.qexp_t(lq).numpy()
    pa["qlog_qexp"] = P.qlog_t(P.qexp_t(lq)).numpy()
    np.savez_compressed(os.path.join(HERE, "pose_algebra.npz"), **pa)

    # --- network: reference PoseNet/MapNet modules over the restated ResNet-34 ----------------
    torch.manual_seed(7)
    net = ns.posenet.MapNet(ns.posenet.PoseNet(oracle.resnet34(), droprate=0.0, pretrained=False))
    crit = C.MapNetCriterion(sax=0.0, saq=-3.0, srx=0.0, srq=-3.0, learn_beta=True, learn_gamma=True)
    x, t = oracle.make_batch("mapnet", 2, 64, 85, seed=7)
    params = [{"params": net.parameters()}, {"params": [crit.sax, crit.saq]}, {"params": [crit.srx, crit.srq]}]
    opt = oracle.Optimizer(params, "adam", base_lr=1e-4, weight_decay=5e-4)
    net.train()
    out = {"seed": np.int64(7), "shape": np.array([2, 3, 64, 85])}
    for step in range(2):
        loss, poses = oracle.step_feedfwd(x, net, False, t, crit, opt, train=True)
        out["loss%d" % step] = np.float64(loss)
        out["poses%d" % step] = poses.detach().numpy()
        if step == 0:
            out["gradnorm0"] = np.array([float(p.grad.norm()) for p in net.parameters()])
            out["crit_grad0"] = np.array([float(c.grad) for c in (crit.sax, crit.saq, crit.srx, crit.srq)])
    out["crit_after"] = np.array([float(c) for c in (crit.sax, crit.saq, crit.srx, crit.srq)])
    out["bn1_running_mean"] = net.mapnet.feature_extractor.bn1.running_mean.numpy()
    out["bn1_running_var"] = net.mapnet.feature_extractor.bn1.running_var.numpy()
    net.eval()
    with torch.no_grad():
        out["poses_eval"] = net(x).numpy()
    np.savez_compressed(os.path.join(HERE, "mapnet_tiny.npz"), **out)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
