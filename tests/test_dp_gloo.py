"""Data-parallel path on CPU: two processes over gloo run the staged training step (emulated kernels)
and all-reduce gradient buckets exactly as the RCCL path does (geomapnet_amd/dp.py)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, same_data, out_dir, defer=0):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MAPNET_EMU_THREADS"] = "4"
    os.environ["MN_DP_DEFER"] = str(defer)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    import emu_lib
    import geomapnet_amd as G
    import oracle
    lib = emu_lib.load()
    G.set_compute_dtype("fp32")
    torch.manual_seed(7)
    net = G.MapNet(G.PoseNet(G.resnet34(_binding=lib), droprate=0.0, pretrained=False, _binding=lib))
    crit = G.MapNetCriterion(sax=0.0, saq=-3.0, srx=0.0, srq=-3.0, learn_beta=True, learn_gamma=True, _binding=lib)
    opt = G.Optimizer([{"params": net.parameters()}, {"params": [crit.sax, crit.saq]}, {"params": [crit.srx, crit.srq]}],
                      "adam", base_lr=1e-4, weight_decay=5e-4)
    net.train()
    x, t = oracle.make_batch("mapnet", 1, 40, 53, seed=7 if same_data else 7 + rank)
    loss, _ = G.step_feedfwd(x, net, False, t, crit, opt, True)
    eng = net.mapnet._engine
    torch.save({"loss": loss, "grads": eng.grads().clone(), "params": eng.params.clone()},
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.destroy_process_group()


def _fp16_worker(rank, world, port, out_dir, dtype="fp16", steps=3, transport="fp32"):
    """`steps` staged steps in a loss-scaled mode (fp16: loss scaling, overflow guard, fp16 kernels incl. the fused weight gradient;
    fp16x2m: the split-operand forward pass in front of them) on rank-specific windows; every rank saves its replica after each step"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MAPNET_EMU_THREADS"] = "4"
    os.environ["MN_DP_GRAD_DTYPE"] = transport
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    import emu_lib
    import geomapnet_amd as G
    import oracle
    lib = emu_lib.load()
    G.set_compute_dtype(dtype)
    torch.manual_seed(7)
    net = G.MapNet(G.PoseNet(G.resnet34(_binding=lib), droprate=0.0, pretrained=False, _binding=lib))
    crit = G.MapNetCriterion(sax=0.0, saq=-3.0, srx=0.0, srq=-3.0, learn_beta=True, learn_gamma=True, _binding=lib)
    opt = G.Optimizer([{"params": net.parameters()}, {"params": [crit.sax, crit.saq]}, {"params": [crit.srx, crit.srq]}],
                      "adam", base_lr=1e-4, weight_decay=5e-4)
    net.train()
    eng = net.mapnet._engine
    snaps, losses = [], []
    for step in range(steps):
        x, t = oracle.make_batch("mapnet", 1, 32, 40, seed=100 * step + rank)
        loss, _ = G.step_feedfwd(x, net, False, t, crit, opt, True)
        losses.append(loss)
        snaps.append(eng.params.clone())
    torch.save({"losses": losses, "params": snaps, "state": eng.loss_scale_state(), "grads": eng.grads().clone()},
               os.path.join(out_dir, "fp16_rank%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.slow
@pytest.mark.parametrize("dtype,steps", [("fp16", 3), ("fp16x2m", 2)])
def test_two_rank_fp16_replicas_stay_bit_identical_over_three_steps(tmp_path, dtype, steps):
    """the data-parallel step of the loss-scaled modes (fp16; fp16x2m, the scripts' default since round 5): gradients differ per rank
    before the all-reduce (different windows), parameters must be bit-identical on both ranks after every optimiser step, the reported
    loss is the same mean on both, and no step was skipped"""
    port = 33500 + os.getpid() % 2000 + (0 if dtype == "fp16" else 2000)
    mp.spawn(_fp16_worker, args=(2, port, str(tmp_path), dtype, steps), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "fp16_rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "fp16_rank1.pt"))
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)
    assert not torch.equal(r0["params"][0], r0["params"][-1])  # the steps did move the weights
    assert r0["losses"] == r1["losses"] and all(l == l for l in r0["losses"])
    assert r0["state"] == r1["state"] and r0["state"][1] == 0


@pytest.mark.slow
def test_two_rank_bf16_gradient_transport(tmp_path):
    """MN_DP_GRAD_DTYPE=bf16 (geomapnet_amd/dp.py; mn_grad_bucket_pack_bf16 / _unpack_bf16): the buckets travel as bf16.  Replicas stay
    bit-identical (every rank widens the same all-reduced values), no step is skipped, no NaN (fp16 halves of the loss-scaled
    gradients overflowed here: weight gradients reach O(100)), and the all-reduced gradient equals the fp32-transported one to bf16's
    rounding (2^-9 per rank)"""
    port = 37500 + os.getpid() % 2000
    mp.spawn(_fp16_worker, args=(2, port, str(tmp_path), "fp16", 1, "bf16"), nprocs=2, join=True)
    h0 = torch.load(os.path.join(tmp_path, "fp16_rank0.pt"))
    h1 = torch.load(os.path.join(tmp_path, "fp16_rank1.pt"))
    assert torch.isfinite(h0["grads"]).all()
    assert torch.equal(h0["params"][0], h1["params"][0]) and torch.equal(h0["grads"], h1["grads"])
    assert h0["losses"] == h1["losses"] and h0["state"][1] == 0
    mp.spawn(_fp16_worker, args=(2, port + 1, str(tmp_path), "fp16", 1, "fp32"), nprocs=2, join=True)
    f0 = torch.load(os.path.join(tmp_path, "fp16_rank0.pt"))
    rel = ((h0["grads"] - f0["grads"]).norm() / f0["grads"].norm()).item()
    assert 0.0 < rel < 4e-3, rel  # rounded (not the fp32 path run twice), and by no more than bf16's rounding
    assert h0["losses"] == f0["losses"]  # the forward pass does not depend on the transport


def _single(seed, out):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_lib
    import geomapnet_amd as G
    import oracle
    lib = emu_lib.load()
    G.set_compute_dtype("fp32")
    torch.manual_seed(7)
    net = G.MapNet(G.PoseNet(G.resnet34(_binding=lib), droprate=0.0, pretrained=False, _binding=lib))
    crit = G.MapNetCriterion(sax=0.0, saq=-3.0, srx=0.0, srq=-3.0, learn_beta=True, learn_gamma=True, _binding=lib)
    opt = G.Optimizer([{"params": net.parameters()}, {"params": [crit.sax, crit.saq]}, {"params": [crit.srx, crit.srq]}],
                      "adam", base_lr=1e-4, weight_decay=5e-4)
    net.train()
    x, t = oracle.make_batch("mapnet", 1, 40, 53, seed=seed)
    loss, _ = G.step_feedfwd(x, net, False, t, crit, opt, True)
    eng = net.mapnet._engine
    return loss, eng.grads().clone(), eng.params.clone()


@pytest.mark.slow
def test_two_rank_gradient_allreduce_matches_single_process(tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, False, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    # replicas stay identical
    assert torch.equal(r0["params"], r1["params"])
    assert torch.equal(r0["grads"], r1["grads"])
    l0, g0, _ = _single(7, None)
    l1, g1, _ = _single(8, None)
    # all-reduced buckets = sum of the per-rank gradients; reported loss = mean of the rank losses
    want = g0 + g1
    assert (r0["grads"] - want).abs().max().item() <= 1e-5 * want.abs().max().item()
    assert abs(r0["loss"] - 0.5 * (l0 + l1)) <= 1e-5 * abs(l0)


@pytest.mark.slow
@pytest.mark.parametrize("defer", [1, 2])
def test_two_rank_deferred_bucket_schedules_give_the_same_step(tmp_path, defer):
    """MN_DP_DEFER (geomapnet_amd/dp.py): bucket 3 at once and buckets 2..0 after the last backward stage (1), or every bucket
    after the last stage (2) -- schedules for when RCCL's workgroups must not sit beside the one-round convolution launches
    (profiles/r06/rccl_rehearsal.txt).  WHEN a bucket is reduced must not change WHAT the step computes: all-reduced gradients =
    sum of the per-rank gradients, replicas identical, as under the default schedule"""
    port = 35500 + os.getpid() % 2000 + 100 * defer
    mp.spawn(_worker, args=(2, port, False, str(tmp_path), defer), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    assert torch.equal(r0["params"], r1["params"]) and torch.equal(r0["grads"], r1["grads"])
    l0, g0, _ = _single(7, None)
    l1, g1, _ = _single(8, None)
    want = g0 + g1
    assert (r0["grads"] - want).abs().max().item() <= 1e-5 * want.abs().max().item()
    assert abs(r0["loss"] - 0.5 * (l0 + l1)) <= 1e-5 * abs(l0)


def _train_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    os.environ["MAPNET_EMU_THREADS"] = "4"
    import configparser
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    import emu_lib
    import train as train_script
    s = configparser.ConfigParser()
    s.read(os.path.join(ROOT, "scripts", "configs", "synthetic_mapnet.ini"))
    s["training"].update(n_epochs="1", batch_size="1", snapshot="1", do_val="no", num_workers="0")
    s["hyperparameters"]["skip"] = "1"
    cfg = os.path.join(out_dir, "mapnet_rank%d.ini" % rank)
    with open(cfg, "w") as f:
        s.write(f)
    args = train_script.build_parser().parse_args(
        ["--model", "mapnet", "--config_file", cfg, "--learn_beta", "--learn_gamma", "--dtype", "fp32", "--synthetic_length", "2",
         "--height", "32", "--width", "40", "--logdir", os.path.join(out_dir, "logs")])
    lines = []
    tr = train_script.run(args, _binding=emu_lib.load(), log=lines.append)
    eng = tr.model.mapnet._engine
    torch.save({"params": eng.params.clone(), "lines": lines, "indices": list(iter(tr.train_sampler)),
                "batches": len(tr.train_loader)}, os.path.join(out_dir, "train_rank%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.slow
def test_two_rank_trainer_shards_windows_and_keeps_replicas_in_sync(tmp_path):
    """scripts/train.py under a 2-rank process group: DistributedSampler shards the windows, every step all-reduces
    the gradient buckets (replicas end identical), rank 0 alone prints and writes the checkpoints"""
    port = 31500 + os.getpid() % 2000
    mp.spawn(_train_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "train_rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "train_rank1.pt"))
    assert torch.equal(r0["params"], r1["params"])
    assert r0["batches"] == 1 and r1["batches"] == 1
    assert sorted(r0["indices"] + r1["indices"]) == [0, 1]
    assert any(l.startswith("Train ") for l in r0["lines"]) and not r1["lines"]
    ck = torch.load(os.path.join(tmp_path, "logs", "epoch_001.pth.tar"), weights_only=False)
    assert ck["epoch"] == 1 and ck["optim_state_dict"]["state"][0]["step"] == 1
