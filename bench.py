#!/usr/bin/env python
"""Benchmark of the MapNet training hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): MapNet,
ResNet-34, 256x341, window T=3, 64 windows = 192 images per GPU per step, fp16 operands / fp32
accumulate, MapNetCriterion with learned beta/gamma, Adam lr 1e-4 wd 5e-4; synthetic inputs
resident in HBM before the timed region; random-init weights.  One "step" = one call of
geomapnet_amd.step_feedfwd(train=True) = forward + criterion + backward + Adam, including the
blocking loss read-back the reference performs (common/train.py:361).  N>1: windows are sharded,
one process per GPU, gradient buckets all-reduced over RCCL while backward continues (weak scaling).

Prints ONE JSON line on rank 0 with `roofline` (all conv MFMA launches of a step, timed with HIP
event pairs on the launch stream inside the timed region) and `cpu_baseline` (the oracle, a port of
the reference path, timed on this host on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_IMAGE_TRAIN = 39.06   # 3 x 13.02 GFLOP (fwd + dgrad + wgrad), conv + linear
GFLOP_PER_IMAGE_EXECUTED = 38.65  # the stem's unused input gradient is not computed
PEAK_F16_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense fp16/bf16 MFMA
PEAK_F32_TFLOPS = 157.3


def cpu_baseline(windows, steps, H, W):
    """oracle (port of the reference CPU path) on this host: MapNet step, fp32, all cores"""
    import oracle
    torch.manual_seed(7)
    net = oracle.MapNet(oracle.PoseNet(oracle.resnet34(), droprate=0.0, pretrained=False))
    crit = oracle.MapNetCriterion(0.0, -3.0, 0.0, -3.0, True, True)
    opt = oracle.Optimizer([{"params": net.parameters()}, {"params": [crit.sax, crit.saq]},
                            {"params": [crit.srx, crit.srq]}], "adam", base_lr=1e-4, weight_decay=5e-4)
    x, t = oracle.make_batch("mapnet", windows, H, W, seed=7)
    net.train()
    oracle.step_feedfwd(x, net, False, t, crit, opt, True)  # warm-up
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        oracle.step_feedfwd(x, net, False, t, crit, opt, True)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    return {"value": round(windows * 3 / med, 2), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle MapNet train step (fwd+loss+bwd+Adam), fp32, %d windows x T=3 = %d images %dx%d, "
                      "1 warm-up + %d timed steps, median %.3f s/step" % (windows, windows * 3, H, W, steps, med)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--windows", type=int, default=64, help="windows per GPU per step")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=341)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="do not time conv launches with HIP events")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import geomapnet_amd as G
    from geomapnet_amd.posenet import engine_of
    G.set_compute_dtype(args.dtype)
    torch.manual_seed(7)
    net = G.MapNet(G.PoseNet(G.resnet34(), droprate=0.0, pretrained=False)).cuda()
    crit = G.MapNetCriterion(sax=0.0, saq=-3.0, srx=0.0, srq=-3.0, learn_beta=True, learn_gamma=True).cuda()
    opt = G.Optimizer([{"params": net.parameters()}, {"params": [crit.sax, crit.saq]}, {"params": [crit.srx, crit.srq]}],
                      "adam", base_lr=1e-4, weight_decay=5e-4)
    net.train()
    n, T, H, W = args.windows, 3, args.height, args.width
    gen = torch.Generator(device=dev).manual_seed(7 + rank)
    images = torch.randn(n, T, 3, H, W, device=dev, generator=gen)
    trans = torch.randn(n, T, 3, device=dev, generator=gen)
    axis = torch.randn(n, T, 3, device=dev, generator=gen)
    axis = axis / axis.norm(dim=-1, keepdim=True)
    half = 0.05 + 1.15 * torch.rand(n, T, 1, device=dev, generator=gen)
    targets = torch.cat((trans, axis * half), dim=-1).contiguous()

    eng = engine_of(net)
    losses = []
    for _ in range(args.warmup):
        l, _ = G.step_feedfwd(images, net, True, targets, crit, opt, True)
        losses.append(l)
    plan = next(iter(eng.plans.values()))
    use_events = not args.no_events

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import ctypes as C
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        l, _ = G.step_feedfwd(images, net, True, targets, crit, opt, True)
        losses.append(l)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()

    # Kernel-level roofline: the timed steps above replay a captured hipGraph (no room for event records),
    # so the conv MFMA launches are timed with HIP event pairs on the launch stream over a second run of the
    # SAME K steps issued eagerly; kernel durations do not depend on how the launch was issued.
    conv_ms, conv_launches, eager_ms = 0.0, 0, None
    if use_events:  # every rank takes part: with world > 1 each step contains collectives
        eng.lib.check(eng.lib.set_profiling(plan["handle"], 1))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            G.step_feedfwd(images, net, True, targets, crit, opt, True)
            ms, cnt = C.c_float(), C.c_int()
            eng.lib.check(eng.lib.last_kernel_ms(plan["handle"], 0, C.byref(ms), C.byref(cnt)))
            conv_ms += ms.value
            conv_launches += cnt.value
        torch.cuda.synchronize()
        eager_ms = 1e3 * (time.perf_counter() - t1) / args.steps
        eng.lib.check(eng.lib.set_profiling(plan["handle"], 0))
    if world > 1:
        dist.barrier()

    if rank == 0:
        images_per_step = n * T * world
        ms_per_step = 1e3 * elapsed / args.steps
        value = images_per_step / (elapsed / args.steps)
        peak = PEAK_F16_TFLOPS if args.dtype == "fp16" else PEAK_F32_TFLOPS
        roof = None
        if use_events and conv_ms > 0:
            conv_ms_step = conv_ms / args.steps
            ach = GFLOP_PER_IMAGE_TRAIN * n * T / conv_ms_step  # GFLOP / ms = TFLOP/s, per GPU
            traffic = None  # HBM bytes of the same launches, from separate rocprofv3 --pmc passes (profiles/)
            try:
                with open(os.path.join(ROOT, "profiles", "r01", "pmc_conv_traffic.json")) as f:
                    traffic = json.load(f)["hbm_bytes_per_step"] if (args.dtype == "fp16" and n == 64) else None
            except Exception:
                pass
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": traffic, "kernel": "igemm_kernel + conv_halo_kernel + wgrad_kernel (all %d conv launches of a step)" % (conv_launches // args.steps),
                    "conv_ms_per_step": round(conv_ms_step, 3), "eager_profiled_ms_per_step": round(eager_ms, 3),
                    "flops_per_step_G": round(GFLOP_PER_IMAGE_TRAIN * n * T, 1),
                    "whole_step_frac": round(GFLOP_PER_IMAGE_TRAIN * n * T / ms_per_step / peak, 4)}
        out = {"metric": "images/sec MapNet ResNet-34 256x341 T=3 train step", "value": round(value, 2), "unit": "images/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f16" if args.dtype == "fp16" else "f32", "data": "synthetic",
               "config": {"workload": "BASELINE configs[2]: MapNet ResNet-34, %d windows x T=3 = %d images/GPU/step, %dx%d, "
                                      "MapNetCriterion learned beta/gamma, Adam, random-init weights" % (n, n * T, H, W),
                          "global_windows": n * world, "parallelism": "dp%d" % world, "loss_first": round(losses[0], 4),
                          "loss_last": round(losses[-1], 4), "gflop_per_image": GFLOP_PER_IMAGE_TRAIN,
                          "gflop_per_image_executed": GFLOP_PER_IMAGE_EXECUTED},
               "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(5, 5, H, W)
            except Exception as e:  # the baseline must never hide the GPU number
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
