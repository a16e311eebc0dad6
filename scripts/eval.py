#!/usr/bin/env python
"""Evaluation script for PoseNet and the MapNet variants on the MI355X forward path.

Command line of the reference's scripts/eval.py (/root/reference/scripts/eval.py:29-45: --dataset --scene
--weights --model --device --config_file --val --output_dir --pose_graph) and its flow: load the checkpoint's
`model_state_dict` through the prefix-aware `load_state_dict`, run windows of `steps` frames with batch size 1,
keep the middle prediction, optionally optimise every window's pose graph (`--pose_graph`, one batched HIP
launch over all windows), un-normalise, report median / mean translation and rotation error
(eval.py:153-205).  Additions: `--dataset Synthetic` (+ `--synthetic_length --height --width`), `--dtype`;
`--output_dir` writes the predicted and target poses as a .npz instead of the reference's matplotlib figure
and pickle.
"""
import argparse
import configparser
import os
import os.path as osp
import sys

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build_parser():
    parser = argparse.ArgumentParser(description="Evaluation script for PoseNet and MapNet variants")
    parser.add_argument("--dataset", type=str, choices=("7Scenes", "RobotCar", "Synthetic"), default="Synthetic",
                        help="Dataset")
    parser.add_argument("--scene", type=str, default="synthetic", help="Scene name")
    parser.add_argument("--weights", type=str, help="trained weights to load")
    parser.add_argument("--model", choices=("posenet", "mapnet", "mapnet++"),
                        help="Model to use (mapnet covers MapNet and MapNet++: they differ only in the weights file)")
    parser.add_argument("--device", type=str, default="0", help="GPU device(s)")
    parser.add_argument("--config_file", type=str, help="configuration file")
    parser.add_argument("--val", action="store_true", help="evaluate the validation split")
    parser.add_argument("--output_dir", type=str, default=None, help="Output directory")
    parser.add_argument("--pose_graph", action="store_true", help="Turn on Pose Graph Optimization")
    # additions
    parser.add_argument("--dropout_active", action="store_true",
                        help="apply the config's dropout on the device in training mode (default: identity, the reference's "
                             "behaviour under its pinned PyTorch 0.4.1)")
    parser.add_argument("--dtype", choices=("fp16", "fp16x2m", "fp16x2q", "fp16x2", "fp32x3", "fp32"), default="fp16x2m",
                        help="compute precision (default: the split-fp16 forward pass, whose poses match the fp32 reference to 1e-3; "
                             "fp16x2q is experimental: fp8 cross terms that saturate silently beyond |activation| 448 / |weight| 7)")
    parser.add_argument("--synthetic_length", type=int, default=256)
    parser.add_argument("--u8_input", action="store_true", help="frames as uint8 [H,W,3]; ToTensor + Normalize run on the "
                        "device (model.set_input_u8)")
    parser.add_argument("--height", type=int, default=256)
    parser.add_argument("--width", type=int, default=341)
    return parser


def run(args, dataset=None, pose_stats=None, _binding=None, log=print):
    """returns (summary dict, pred_poses [L,7], targ_poses [L,7])"""
    import numpy as np
    import torch
    from torch.utils.data import DataLoader
    import geomapnet_amd as G
    from geomapnet_amd import evaluate as E
    from geomapnet_amd.data import MF, SyntheticFrames, calc_vos_safe, calc_vos_safe_fc

    if "CUDA_VISIBLE_DEVICES" not in os.environ:
        os.environ["CUDA_VISIBLE_DEVICES"] = args.device
    G.set_compute_dtype(args.dtype)
    kw = {} if _binding is None else {"_binding": _binding}

    settings = configparser.ConfigParser()
    with open(args.config_file, "r") as f:
        settings.read_file(f)
    seed = settings.getint("training", "seed")
    section = settings["hyperparameters"]
    dropout = section.getfloat("dropout")
    windows = (args.model.find("mapnet") >= 0) or args.pose_graph
    sig = {}
    if windows:
        steps = section.getint("steps")
        skip = section.getint("skip")
        real = section.getboolean("real")
        variable_skip = section.getboolean("variable_skip")
        fc_vos = args.dataset == "RobotCar"
        if args.pose_graph:
            sig = dict(sax=section.getfloat("s_abs_trans", 1), saq=section.getfloat("s_abs_rot", 1),
                       srx=section.getfloat("s_rel_trans", 20), srq=section.getfloat("s_rel_rot", 20))

    # model
    feature_extractor = G.resnet34(pretrained=False, **kw)
    posenet = G.PoseNet(feature_extractor, droprate=dropout, pretrained=False, dropout_active=args.dropout_active, **kw)
    if args.model.find("mapnet") >= 0:
        model = G.MapNet(mapnet=posenet)
    else:
        model = posenet
    model.eval()
    if args.u8_input:
        model.set_input_u8(SyntheticFrames.MEAN, SyntheticFrames.STD)

    # load weights
    weights_filename = osp.expanduser(args.weights)
    if not osp.isfile(weights_filename):
        log("Could not load weights from {:s}".format(weights_filename))
        sys.exit(-1)
    checkpoint = torch.load(weights_filename, map_location=lambda storage, loc: storage, weights_only=False)
    G.load_state_dict(model, checkpoint["model_state_dict"])
    log("Loaded weights from {:s}".format(weights_filename))

    # mean and stdev for un-normalising the predictions (the scene's pose_stats.txt in the reference)
    pose_m, pose_s = pose_stats if pose_stats is not None else (np.zeros(3), np.ones(3))

    # dataset
    train = not args.val
    log("Running {:s} on {:s} data".format(args.model, "TRAIN" if train else "VAL"))
    if dataset is None:
        if args.dataset != "Synthetic":
            raise NotImplementedError(
                "the {:s} image reader is host-side file parsing outside the MI355X hot path: pass the frame dataset "
                "to run(args, dataset=...), or use --dataset Synthetic".format(args.dataset))
        dataset = SyntheticFrames(args.synthetic_length, H=args.height, W=args.width, seed=seed + (0 if train else 1),
                                  uint8=args.u8_input)
    if windows:
        if args.pose_graph:
            assert real
        vo_func = calc_vos_safe_fc if fc_vos else calc_vos_safe
        data_set = MF(dataset, steps=steps, skip=skip, real=real, variable_skip=variable_skip,
                      include_vos=args.pose_graph, vo_func=vo_func, no_duplicates=False, train=train,
                      gt_dataset=dataset if real else None)
    else:
        data_set = dataset

    # loader (batch_size MUST be 1)
    loader = DataLoader(data_set, batch_size=1, shuffle=False, num_workers=0, pin_memory=torch.cuda.is_available())

    CUDA = torch.cuda.is_available()
    torch.manual_seed(seed)
    if CUDA:
        model.cuda()

    # eval.py:158-163: with --pose_graph the middle prediction of a window goes to the row of ITS frame
    scatter = data_set.get_indices if (args.pose_graph and hasattr(data_set, "get_indices")) else None
    summary, pred_poses, targ_poses = E.evaluate(model, loader, pose_m, pose_s, cuda=CUDA, pose_graph=args.pose_graph,
                                                 fc_vos=fc_vos if windows else False, indices_of=scatter,
                                                 length=len(data_set) if scatter else None, **sig)
    log("Error in translation: median {:3.2f} m,  mean {:3.2f} m\n"
        "Error in rotation: median {:3.2f} degrees, mean {:3.2f} degree".format(
            summary["median_t"], summary["mean_t"], summary["median_q"], summary["mean_q"]))
    if args.output_dir is not None:
        os.makedirs(osp.expanduser(args.output_dir), exist_ok=True)
        name = "{:s}_{:s}_{:s}{:s}.npz".format(args.dataset, args.scene, args.model, "_pgo" if args.pose_graph else "")
        fn = osp.join(osp.expanduser(args.output_dir), name)
        np.savez(fn, pred_poses=pred_poses, targ_poses=targ_poses, **summary)
        log("{:s} saved".format(fn))
    return summary, pred_poses, targ_poses


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.config_file is None or args.model is None or args.weights is None:
        build_parser().error("--config_file, --model and --weights are required")
    run(args)


if __name__ == "__main__":
    main()
