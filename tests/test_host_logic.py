"""CPU tests of the host side: C-ABI library loads and exports every declared symbol, model layout /
state_dict surface matches the reference, the product never touches the oracle."""
import ctypes
import os
import re

import pytest
import torch

import oracle
from geomapnet_amd import _binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hip_library_exports_every_declared_symbol():
    """no compute calls (no GPU here): dlopen + symbol table only"""
    assert os.path.isfile(_binding.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_binding.LIB_PATH)
    hdr = open(os.path.join(ROOT, "include", "mapnet_hip.h")).read()
    declared = set(re.findall(r"\b(mn_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"mn_handle"}
    assert declared == set(_binding.SYMBOLS), declared ^ set(_binding.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    lib.mn_backend.restype = ctypes.c_char_p
    assert lib.mn_backend() == b"hip"


def test_product_fails_loudly_without_gpu():
    import geomapnet_amd as G
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    net = G.PoseNet(G.resnet34(), droprate=0.0, pretrained=False)
    with pytest.raises(_binding.MapNetHipError):
        net(torch.zeros(1, 3, 64, 64))
    with pytest.raises(_binding.MapNetHipError):
        G.MapNetCriterion()(torch.zeros(2, 3, 6), torch.zeros(2, 3, 6))


def test_product_never_imports_the_oracle_or_the_emulator():
    bad = []
    for top in ("geomapnet_amd", "scripts"):  # the package and the command lines on top of it
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if not f.endswith((".py", ".h", ".hip", ".cpp")):
                    continue
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, re.M) or "libmapnet_emu" in src or "emu_lib" in src:
                    bad.append(f)
    assert not bad, bad


def test_state_dict_surface_matches_reference_layout():
    import emu_lib
    import geomapnet_amd as G
    lib = emu_lib.load()
    torch.manual_seed(3)
    onet = oracle.MapNet(oracle.PoseNet(oracle.resnet34(), droprate=0.0, pretrained=False))
    net = G.MapNet(G.PoseNet(G.resnet34(_binding=lib), droprate=0.5, pretrained=False, _binding=lib))
    osd, sd = onet.state_dict(), net.state_dict()
    assert list(sd.keys()) == list(osd.keys())
    assert [tuple(v.shape) for v in sd.values()] == [tuple(v.shape) for v in osd.values()]
    assert [k for k, _ in net.named_parameters()] == [k for k, _ in onet.named_parameters()]
    assert sum(p.numel() for p in net.parameters()) == 22347590
    # round trip through the OHWI arena
    net.load_state_dict(osd)
    for (k, a), b in zip(net.state_dict().items(), osd.values()):
        assert torch.equal(a, b), k
    # common/train.py:29-42 compares the first parameter name of model and checkpoint
    assert next(iter(net.named_parameters()))[0] == "mapnet.feature_extractor.conv1.weight"
    # PoseNet init statistics (kaiming_normal_, zero biases) -- models/posenet.py:59-63
    pn = net.mapnet
    w = dict(pn.named_parameters())["feature_extractor.layer2.0.conv1.weight"]
    assert abs(w.std().item() - (2.0 / (64 * 9)) ** 0.5) < 0.05 * (2.0 / (64 * 9)) ** 0.5
    assert float(pn.fc_xyz.bias.abs().max()) == 0.0


def test_dropout_identity_warns_and_active_dropout_does_not():
    """PoseNet(droprate > 0) is an identity (the reference under its pinned PyTorch 0.4.1) and says so loudly;
    dropout_active=True arms the device operator instead (mn_set_dropout) and is silent"""
    import warnings
    import emu_lib
    import geomapnet_amd as G
    lib = emu_lib.load()
    with pytest.warns(UserWarning, match="IDENTITY"):
        net = G.PoseNet(G.resnet34(_binding=lib), droprate=0.5, pretrained=False, _binding=lib)
    assert net._engine.dropout == (0.0, 0)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        net = G.PoseNet(G.resnet34(_binding=lib), droprate=0.5, pretrained=False, dropout_active=True, dropout_seed=5, _binding=lib)
        G.PoseNet(G.resnet34(_binding=lib), droprate=0.0, pretrained=False, _binding=lib)
    assert net._engine.dropout == (0.5, 5)


def test_criterion_facade_surface():
    import geomapnet_amd as G
    c = G.MapNetCriterion(sax=0.0, saq=-3.0, srx=0.0, srq=-3.0, learn_beta=True, learn_gamma=False)
    assert list(c.state_dict().keys()) == ["sax", "saq", "srx", "srq"]
    assert c.sax.requires_grad and not c.srx.requires_grad and tuple(c.saq.shape) == (1,)
    assert float(c.saq) == -3.0
    p = G.PoseNetCriterion(saq=-3.0)
    assert list(p.state_dict().keys()) == ["sax", "saq"]
    o = G.MapNetOnlineCriterion(gps_mode=True)
    assert o.mode == 3 and G.MapNetOnlineCriterion().mode == 2


def test_evaluate_metric_matches_reference_golden(golden_dir):
    """geomapnet_amd.evaluate (host side of scripts/eval.py) against vectors computed by the reference's own
    numpy helpers: qexp, un-normalisation, L2 translation error, quaternion angular error, median / mean"""
    import numpy as np
    from geomapnet_amd import evaluate as E
    g = np.load(os.path.join(golden_dir, "eval_metric.npz"))
    pred7 = E.to_pose7(g["pred"], g["pose_m"], g["pose_s"])
    targ7 = E.to_pose7(g["targ"], g["pose_m"], g["pose_s"])
    np.testing.assert_allclose(pred7, g["pred7"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(targ7, g["targ7"], rtol=0, atol=1e-13)
    t_loss, q_loss = E.pose_errors(pred7, targ7)
    np.testing.assert_allclose(t_loss, g["t_loss"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(q_loss, g["q_loss"], rtol=0, atol=1e-10)
    s = E.summarize(t_loss, q_loss)
    np.testing.assert_allclose([s["median_t"], s["mean_t"], s["median_q"], s["mean_q"]], g["stats"], rtol=1e-12)
    lq = [E.log_quaternion_angular_error(p[3:], t[3:]) for p, t in zip(g["pred"], g["targ"])]
    np.testing.assert_allclose(lq, g["logq_err"], rtol=0, atol=1e-10)
    assert t_loss[0] == 0.0 and q_loss[0] < 1e-5           # identical pose
    assert 0.0 <= q_loss[2] <= 180.0                       # opposite rotation vector


def test_batch_construction_matches_reference_golden(golden_dir):
    """geomapnet_amd.data (MF window indices, VO targets) against vectors produced by the reference's own
    MF.get_indices / calc_vos_safe / calc_vos_simple"""
    import numpy as np
    from geomapnet_amd import data as D
    g = np.load(os.path.join(golden_dir, "batch_construction.npz"))
    for row in g["get_indices"]:
        steps, skip, nodup, L, index = [int(v) for v in row[:5]]
        mf = D.MF(list(range(L)), steps=steps, skip=skip, no_duplicates=bool(nodup))
        np.testing.assert_array_equal(mf.get_indices(index), row[5:5 + steps])
        assert len(mf) == L - ((steps - 1) * skip if nodup else 0)
    poses = torch.from_numpy(g["poses"])
    np.testing.assert_allclose(D.calc_vos_safe(poses).numpy(), g["vos_safe"], rtol=0, atol=2e-6)
    np.testing.assert_array_equal(D.calc_vos_simple(poses).numpy(), g["vos_simple"])


def test_mf_and_mfonline_batches_feed_the_criteria_shapes():
    """MF / MFOnline over a synthetic frame dataset produce the [N,T,...] batches step_feedfwd expects
    (composite.py:76-126): MapNet [T,3,H,W] + [T,6]; MapNet++ [2T,3,H,W] + [T + (T-1), 6]; gps [2T,6]"""
    import numpy as np
    from geomapnet_amd import data as D
    ds = D.SyntheticFrames(20, 8, 9)
    mf = D.MF(ds, steps=3, skip=10, include_vos=False)
    ims, poses = mf[10]
    assert ims.shape == (3, 3, 8, 9) and poses.shape == (3, 6)
    assert torch.equal(poses[1], ds.poses[10])                      # the middle frame is the indexed one
    val = D.SyntheticFrames(30, 8, 9, seed=3)
    on = D.MFOnline(ds, val, val_gt_dataset=val, steps=3, skip=2)
    ims, poses = on[5]
    assert ims.shape == (6, 3, 8, 9) and poses.shape == (5, 6) and len(on) == 30 - 2 * 2
    idx = on.val_set.get_indices(5)
    want = D.calc_vos_safe(val.poses[idx].unsqueeze(0))[0]
    assert torch.equal(poses[3:], want)
    gps = D.MFOnline(ds, val, gps_mode=True, steps=3, skip=2)
    assert gps[5][1].shape == (6, 6)
    # process_poses: identity alignment returns (t - mean)/std and the log-quaternion of R
    q = np.array([0.9, 0.1, -0.3, 0.2]); q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    raw = np.hstack((R, np.array([[1.0], [2.0], [3.0]]))).reshape(1, 12)
    out = D.process_poses(raw, mean_t=np.array([0.5, 0.5, 0.5]), std_t=np.array([2.0, 2.0, 2.0]), align_R=np.eye(3),
                          align_t=np.zeros(3), align_s=1.0)
    np.testing.assert_allclose(out[0, :3], [0.25, 0.75, 1.25])
    np.testing.assert_allclose(out[0, 3:], D.qlog(q), atol=1e-12)


def test_shipped_k_loops_have_no_scratch_access_and_asm_reads_skip_the_dma_wait():
    """static check of the built gfx950 code objects (tools/isa_audit.py): no scratch access around the K loops of the
    fp16 conv kernels the training step launches (a spill reload there waits on vmcnt(0) behind the LDS-DMA queue), and
    the assembly-read weight-gradient variants go from their LDS-DMA issue to the transpose reads without the
    compiler's vmcnt(0)"""
    import importlib.util
    import shutil
    lib = os.path.join(ROOT, "geomapnet_amd", "libmapnet_hip.so")
    if not os.path.isfile(lib) or not os.path.isfile("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("needs the built library and llvm-objdump")
    spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(ROOT, "tools", "isa_audit.py"))
    audit = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(audit)
    # (igemm_kernel<half, WM, WN, TM, TN, NP, NBUF, MINW, UNI, SPL, ABL = 0, MM = 0>: the 12-wave, 128x128 and 256x128
    # configurations; wgrad_dma_kernel<BMO, BNO, 32, 4, FAST = true, ASMRD = true>: four tile shapes)
    hot = audit.signatures(r"igemm_kernelIDF16_Li3ELi4ELi3ELi2ELi8ELi2ELi3ELb1ELb1ELi0ELi0E|"
                           r"igemm_kernelIDF16_Li2ELi2ELi2ELi2ELi8ELi2ELi2ELb1ELb0ELi0ELi0E|"
                           r"igemm_kernelIDF16_Li2ELi2ELi4ELi2ELi4ELi3ELi2ELb1ELb0ELi0ELi0E|"
                           r"wgrad_dma_kernelILi\d+ELi\d+ELi32ELi\dELb1ELb1E|"
                           r"igemm_halo_kernelILi\d+ELi\d+ELi0ELi1ELb[01]ELi\dELi\dELb[01]ELi[12]E")  # igemm_halo_kernel<BN, kAH, ABL = 0, DP = 1, H2, WM, WN, A1, OCC, Q>
    # two fp16 shapes + the four h2 shapes of the fp16x2 mode (256 columns, 128 columns at 288 / 384 rows, layer1's 64 columns)
    # (igemm.h: 3, wgrad_dma: 4, igemm_halo: the fp16 shapes 256 / 128 x 288 / 128 x 384 + the four h2 shapes; every shape that
    # is added joins the audit below by matching the pattern)
    assert len(hot) >= 14, sorted(hot)
    for name, (_, _, sig) in hot.items():
        assert "S!" not in sig, (name, sig)
    asm = {n: s for n, (_, _, s) in hot.items() if "wgrad_dma" in n and n.split("ELb1ELb")[1].startswith("1")}
    assert len(asm) == 4
    for name, sig in asm.items():
        toks = sig.split()
        k = next(i for i, t in enumerate(toks) if re.fullmatch(r"Rx\d+", t) and toks[i + 1].startswith("M"))
        d = max(i for i in range(k) if toks[i].startswith("D"))  # the last LDS-DMA issue before the transpose reads
        assert not any(t.startswith("V") for t in toks[d:k]), (name, sig)


def test_evaluate_scatters_middle_predictions_by_frame_index(monkeypatch):
    """eval.py:158-190: with an index callback the middle prediction of batch i is written at row get_indices(i)[middle]
    of zero-initialised [L, 7] arrays (later windows overwrite earlier ones on the same frame)"""
    import numpy as np
    import torch
    from geomapnet_amd import evaluate as E

    class M:
        training = False

        def eval(self):
            return self

        def train(self, mode=True):
            return self

    outs = [torch.full((1, 3, 6), float(i + 1)) * 0.1 for i in range(4)]
    it = iter(outs)
    monkeypatch.setattr(E, "step_feedfwd", lambda data, model, cuda, train=False: (0, next(it)))
    batches = [(torch.zeros(1, 3, 3, 4, 4), torch.full((1, 3, 6), float(i))) for i in range(4)]
    idx = {0: [0, 0, 1], 1: [0, 1, 2], 2: [1, 2, 3], 3: [2, 2, 3]}  # frames 0 and 2 are the middle of two windows each
    summary, pred, targ = E.evaluate(M(), batches, cuda=False, indices_of=lambda i: idx[i], length=5)
    assert pred.shape == (5, 7) and targ.shape == (5, 7)
    assert np.allclose(pred[0, :3], 0.1) and np.allclose(pred[1, :3], 0.2)
    assert np.allclose(pred[2, :3], 0.4), "the later window overwrites the earlier one"
    assert np.allclose(pred[3], 0) and np.allclose(pred[4], 0)  # no window centres there: rows stay zero, as in the reference
    assert np.allclose(targ[2, :3], 3.0)


def test_device_feed_is_a_pass_through_on_cpu_and_keeps_the_loader_contract():
    """geomapnet_amd.DeviceFeed on a CPU device hands out the loader's own batches, in order, and reports the loader's length"""
    import geomapnet_amd as G
    batches = [(torch.full((2, 3), float(i)), torch.full((2, 6), -float(i))) for i in range(5)]
    feed = G.DeviceFeed(batches, "cpu")
    assert len(feed) == 5
    got = list(feed)
    assert all(a is b for (a, _), (b, _) in zip(got, batches)) and len(got) == 5
    with pytest.raises(ValueError):
        G.DeviceFeed(batches, "cpu", depth=1)


def test_ring_allreduce_duration_model_and_rccl_channel_budget(monkeypatch):
    """dp.py's stand-in duration model and the RCCL environment the data-parallel bench sets before creating its process group"""
    from geomapnet_amd import dp
    assert abs(dp.ring_allreduce_us(57e6, 8, 200.0, 40.0) - (40.0 + 57e6 * 1.75 / 200e3)) < 1e-6
    assert dp.ring_allreduce_us(0.9e6) < dp.ring_allreduce_us(4.5e6) < dp.ring_allreduce_us(27e6)
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "0")  # (so that the teardown restores the variable's original state)
    monkeypatch.delenv("NCCL_MAX_NCHANNELS")
    monkeypatch.setenv("NCCL_MIN_NCHANNELS", "2")  # a user's setting wins
    env = dp.rccl_env()
    assert env["NCCL_MIN_NCHANNELS"] == "2" and int(env["NCCL_MAX_NCHANNELS"]) >= 2
