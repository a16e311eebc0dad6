"""CPU suite: the reference's train.py / eval.py command lines and the Trainer epoch loop
(common/train.py:206-320), run end to end through the SIMT-emulator build of the kernels on a tiny
synthetic sequence: train -> checkpoint -> eval (with and without pose-graph optimisation) -> resume."""
import configparser
import os
import sys

import numpy as np
import pytest
import torch

import emu_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

H, W = 32, 40


@pytest.fixture(scope="module")
def lib():
    return emu_lib.load()


def _config(tmp_path, name, **training):
    """a copy of the shipped config with a short schedule"""
    s = configparser.ConfigParser()
    s.read(os.path.join(ROOT, "scripts", "configs", name))
    for k, v in training.items():
        sect = "hyperparameters" if k in ("skip", "steps", "real") else "training"
        s[sect][k] = str(v)
    fn = str(tmp_path / name)
    with open(fn, "w") as f:
        s.write(f)
    return fn


def _train_args(tmp_path, model, cfg, extra=(), length=2, val_length=1):
    import train as train_script
    argv = ["--model", model, "--config_file", cfg, "--dtype", "fp32", "--synthetic_length", str(length),
            "--synthetic_val_length", str(val_length), "--height", str(H), "--width", str(W), "--logdir",
            str(tmp_path / ("logs_" + model.replace("+", "p"))), "--num_workers", "0"]
    return train_script, train_script.build_parser().parse_args(argv + list(extra))


def test_train_eval_resume_mapnet(lib, tmp_path):
    cfg = _config(tmp_path, "synthetic_mapnet.ini", n_epochs=1, batch_size=2, snapshot=1, val_freq=1, skip=1, do_val="no")
    train_script, args = _train_args(tmp_path, "mapnet", cfg, ["--learn_beta", "--learn_gamma"])
    lines = []
    tr = train_script.run(args, _binding=lib, log=lines.append)
    assert tr.experiment == "Synthetic_synthetic_mapnet_synthetic_mapnet_learn_beta_learn_gamma"
    assert sum(l.startswith("Train ") for l in lines) == 1  # one batch of two windows
    for e in (0, 1):
        assert os.path.isfile(os.path.join(tr.logdir, "epoch_%03d.pth.tar" % e))
    ck = torch.load(tr.final_checkpoint, weights_only=False)
    assert set(ck) == {"epoch", "model_state_dict", "optim_state_dict", "criterion_state_dict"} and ck["epoch"] == 1
    assert set(ck["criterion_state_dict"]) == {"sax", "saq", "srx", "srq"}
    assert float(ck["criterion_state_dict"]["saq"]) != -3.0  # learned
    assert ck["optim_state_dict"]["state"][0]["step"] == 1

    # eval.py flow on the checkpoint
    import eval as eval_script
    ecfg = _config(tmp_path, "synthetic_mapnet.ini", skip=1)
    eargs = eval_script.build_parser().parse_args(
        ["--model", "mapnet", "--config_file", ecfg, "--weights", tr.final_checkpoint, "--dtype", "fp32", "--synthetic_length",
         "1", "--height", str(H), "--width", str(W), "--val", "--output_dir", str(tmp_path / "out")])
    summary, pred, targ = eval_script.run(eargs, _binding=lib, log=lines.append)
    assert pred.shape == (1, 7) and targ.shape == (1, 7) and np.isfinite(pred).all()
    z = np.load(str(tmp_path / "out" / "Synthetic_synthetic_mapnet.npz"))
    np.testing.assert_array_equal(z["pred_poses"], pred)

    # resume: start epoch and optimiser state come from the checkpoint (n_epochs = the checkpoint's epoch: no further
    # step is run here -- the step after a resume is checked in test_emu_network.py's checkpoint test)
    cfg3 = _config(tmp_path, "synthetic_mapnet.ini", n_epochs=1, batch_size=2, snapshot=1, val_freq=5, skip=1, do_val="no")
    _, rargs = _train_args(tmp_path, "mapnet", cfg3, ["--learn_beta", "--learn_gamma", "--checkpoint", tr.final_checkpoint,
                                                      "--resume_optim", "--suffix", "_r"])
    tr2 = train_script.run(rargs, _binding=lib, log=lines.append)
    assert tr2.start_epoch == 1
    ck2 = torch.load(tr2.final_checkpoint, weights_only=False)
    assert ck2["epoch"] == 1 and ck2["optim_state_dict"]["state"][0]["step"] == 1
    for k, v in ck["model_state_dict"].items():
        assert torch.equal(v, ck2["model_state_dict"][k]), k
    assert torch.equal(ck["optim_state_dict"]["state"][0]["exp_avg"], ck2["optim_state_dict"]["state"][0]["exp_avg"])


def test_eval_pose_graph_and_posenet_weights_into_mapnet(lib, tmp_path):
    import geomapnet_amd as G
    import eval as eval_script
    torch.manual_seed(3)
    net = G.PoseNet(G.resnet34(_binding=lib), droprate=0.0, pretrained=False, _binding=lib)
    wfn = str(tmp_path / "w.pth.tar")
    torch.save({"model_state_dict": net.state_dict()}, wfn)  # PoseNet keys: MapNet adds its prefix (common/train.py:22-53)
    cfg = _config(tmp_path, "synthetic_pose_graph.ini", skip=1, steps=3)
    eargs = eval_script.build_parser().parse_args(
        ["--model", "mapnet", "--config_file", cfg, "--weights", wfn, "--dtype", "fp32", "--synthetic_length", "2",
         "--height", str(H), "--width", str(W), "--pose_graph"])
    summary, pred, targ = eval_script.run(eargs, _binding=lib, log=lambda *a: None)
    assert pred.shape == (2, 7) and np.isfinite(pred).all()
    np.testing.assert_allclose(np.linalg.norm(pred[:, 3:], axis=1), 1.0, atol=1e-9)  # optimised poses carry unit quaternions


def test_train_posenet_with_validation_and_mapnet_online_wiring(lib, tmp_path):
    cfg = _config(tmp_path, "synthetic_posenet.ini", n_epochs=1, batch_size=2, snapshot=1, val_freq=1)
    train_script, args = _train_args(tmp_path, "posenet", cfg, ["--learn_beta", "--u8_input"])  # uint8 frames, device-side Normalize
    lines = []
    tr = train_script.run(args, _binding=lib, log=lines.append)
    assert any(l.startswith("Val ") and "val_loss" in l for l in lines) and np.isfinite(tr.last_val_loss)
    data, _ = next(iter(tr.train_loader))
    assert data.dtype == torch.uint8 and tuple(data.shape) == (2, H, W, 3)
    # MapNet++: construction only (n_epochs = 0; the step itself is covered by test_emu_network.py) -- model with the
    # NaN filter, online criterion, MFOnline batches of 2T frames with T + (T-1) target rows
    cfg = _config(tmp_path, "synthetic_mapnet_online.ini", n_epochs=0, batch_size=1, snapshot=1, skip=1)
    _, args = _train_args(tmp_path, "mapnet++", cfg, ["--learn_beta", "--learn_gamma"], length=3, val_length=3)
    tr = train_script.run(args, _binding=lib, log=lambda *a: None)
    data, target = next(iter(tr.train_loader))
    assert tuple(data.shape) == (1, 6, 3, H, W) and tuple(target.shape) == (1, 5, 6)
    assert type(tr.train_criterion).__name__ == "MapNetOnlineCriterion" and tr.config["max_grad_norm"] == 5.0
    assert tr.optimizer.learner.param_groups[0]["lr"] == 1e-5
    assert torch.load(tr.final_checkpoint, weights_only=False)["epoch"] == 0


def test_real_datasets_are_refused(lib, tmp_path):
    cfg = _config(tmp_path, "synthetic_mapnet.ini", n_epochs=1)
    train_script, args = _train_args(tmp_path, "mapnet", cfg, ["--dataset", "7Scenes", "--scene", "chess"])
    with pytest.raises(NotImplementedError):
        train_script.run(args, _binding=lib, log=lambda *a: None)
