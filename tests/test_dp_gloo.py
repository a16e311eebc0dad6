"""Data-parallel path on CPU: two processes over gloo run the staged training step (emulated kernels)
and all-reduce gradient buckets exactly as the RCCL path does (geomapnet_amd/dp.py)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, same_data, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MAPNET_EMU_THREADS"] = "4"
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
