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
    for dirpath, _, files in os.walk(os.path.join(ROOT, "geomapnet_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
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
