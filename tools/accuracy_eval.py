"""Does a precision mode TRAIN to the same accuracy?  (BASELINE.json's metric names "median t/q err".)

A learnable synthetic scene (geomapnet_amd.data.RenderedFrames: the picture is a smooth function of the camera pose) is
trained through the reference's command-line flow -- scripts/train.py `run` -> checkpoint -> scripts/eval.py `run`, i.e.
Trainer.train_val, step_feedfwd, MapNetCriterion with learned beta / gamma, Adam, then the reference's evaluation metric
(median / mean translation and rotation error of the window's middle prediction, scripts/eval.py:153-205) on HELD-OUT frames of
the same scene -- once per dtype, from identical initial weights, data order and seeds.  Prints one JSON line:
{"fp16": {"median_t": .., "median_q": ..}, "fp16x2": {...}, "baseline_predict_mean": {...}}.
usage: python tools/accuracy_eval.py [--dtypes fp16,fp16x2] [--epochs 40] [--train 512] [--val 128] [--height 64] [--width 85]"""
import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "scripts")]


def train_and_eval(dtype, epochs, n_train, n_val, H, W, batch, lr, binding=None, log=lambda *a: None, seed=7, deterministic=False):
    """deterministic: run under MN_DETERMINISTIC=1 (read when a plan is created): every step is bit-reproducible, so a (dtype, seed)
    pair always trains to the same numbers and two dtypes differ by their arithmetic only, not by atomics' summation order"""
    if deterministic:
        old = os.environ.get("MN_DETERMINISTIC")
        os.environ["MN_DETERMINISTIC"] = "1"
        try:
            return train_and_eval(dtype, epochs, n_train, n_val, H, W, batch, lr, binding, log, seed, False)
        finally:
            if old is None:
                os.environ.pop("MN_DETERMINISTIC", None)
            else:
                os.environ["MN_DETERMINISTIC"] = old
    import numpy as np
    import torch
    # the DataLoader's collate (torch.stack of 48 small frames per step) runs on this thread: with the box's 128 cores as intra-op
    # threads each call costs ~1 ms of thread wake-ups (35 ms per 4.6 ms training step, measured); a handful is enough
    torch.set_num_threads(min(8, torch.get_num_threads()))
    import train as train_script
    import eval as eval_script
    from geomapnet_amd.data import RenderedFrames
    cfg = os.path.join(ROOT, "scripts", "configs", "synthetic_mapnet.ini")
    train_frames = RenderedFrames(n_train, H=H, W=W, seed=7, scene_seed=1)
    val_frames = RenderedFrames(n_val, H=H, W=W, seed=8, scene_seed=1)
    with tempfile.TemporaryDirectory() as logdir:
        import configparser
        settings = configparser.ConfigParser()
        settings.read(cfg)
        settings["optimization"]["lr"] = repr(lr)
        settings["training"].update({"n_epochs": str(epochs), "batch_size": str(batch), "num_workers": "0", "do_val": "no",
                                     "snapshot": str(epochs), "print_freq": "1000", "seed": str(seed)})
        ini = os.path.join(logdir, "accuracy.ini")
        with open(ini, "w") as f:
            settings.write(f)
        targs = train_script.build_parser().parse_args(
            ["--dataset", "Synthetic", "--model", "mapnet", "--config_file", ini, "--learn_beta", "--learn_gamma", "--dtype", dtype,
             "--height", str(H), "--width", str(W), "--pretrained", "no", "--logdir", logdir])
        trainer = train_script.run(targs, datasets=(train_frames, val_frames), _binding=binding, log=log)
        ckpt = os.path.join(logdir, "epoch_{:03d}.pth.tar".format(epochs))
        eargs = eval_script.build_parser().parse_args(
            ["--dataset", "Synthetic", "--model", "mapnet", "--config_file", ini, "--weights", ckpt, "--dtype", dtype, "--val",
             "--height", str(H), "--width", str(W)])
        summary, pred, targ = eval_script.run(eargs, dataset=val_frames, _binding=binding, log=log)
        del trainer
    # what predicting the training set's mean pose would score on the same frames (the metric's scale)
    from geomapnet_amd.evaluate import pose_errors, summarize, to_pose7
    mean7 = to_pose7(train_frames.poses.mean(0, keepdim=True).numpy().repeat(len(targ), 0), np.zeros(3), np.ones(3))
    base = summarize(*pose_errors(mean7, np.asarray(targ, dtype=np.float64)))
    return ({k: float(summary[k]) for k in ("median_t", "mean_t", "median_q", "mean_q")},
            {k: float(base[k]) for k in ("median_t", "median_q")})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtypes", default="fp16,fp16x2")
    ap.add_argument("--epochs", type=int, default=40)
    ap.add_argument("--train", type=int, default=512)
    ap.add_argument("--val", type=int, default=128)
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--width", type=int, default=85)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--seeds", default="7", help="comma-separated [training] seeds (initial weights + data order); one run per seed")
    ap.add_argument("--deterministic", action="store_true", help="MN_DETERMINISTIC=1: bit-reproducible steps")
    a = ap.parse_args()
    out = {"config": {"scene": "RenderedFrames scene_seed=1", "train_frames": a.train, "val_frames": a.val, "HxW": [a.height, a.width],
                      "epochs": a.epochs, "windows_per_step": a.batch, "steps": a.epochs * (a.train // a.batch), "lr": a.lr,
                      "deterministic": bool(a.deterministic)}}
    seeds = [int(v) for v in a.seeds.split(",")]
    for d in a.dtypes.split(","):
        runs = []
        for sd in seeds:
            res, base = train_and_eval(d, a.epochs, a.train, a.val, a.height, a.width, a.batch, a.lr, seed=sd,
                                       deterministic=a.deterministic)
            runs.append(dict(res, seed=sd))
            print("# %s seed %d: median_t %.4f median_q %.3f" % (d, sd, res["median_t"], res["median_q"]), file=sys.stderr, flush=True)
        out[d] = runs[0] if len(runs) == 1 else {"runs": runs, "median_t": sum(r["median_t"] for r in runs) / len(runs),
                                                 "median_q": sum(r["median_q"] for r in runs) / len(runs)}
        out["baseline_predict_mean"] = base
    print(json.dumps(out))


if __name__ == "__main__":
    main()
