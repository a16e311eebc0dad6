#!/usr/bin/env python
"""Benchmark of the MapNet training hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher: bench.py starts its own N ranks (python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>), one process per GPU; under an
existing launcher (RANK / WORLD_SIZE in the environment) it is one of the ranks.  Either way the N = 1 code
path and the N > 1 code path are the same function.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): MapNet, ResNet-34, 256x341,
window T=3, 64 windows = 192 images per GPU per step, fp16 operands / fp32 accumulate (--dtype fp32: the exact-fp32
MFMA build the parity bar is met in), MapNetCriterion with learned beta/gamma, Adam lr 1e-4 wd 5e-4; synthetic inputs
resident in HBM before the timed region; random-init weights.  One "step" = one call of
geomapnet_amd.step_feedfwd(train=True) = forward + criterion + backward + Adam, including the blocking loss
read-back the reference performs (common/train.py:361).  N>1: windows are sharded, one process per GPU, gradient
buckets all-reduced over RCCL while backward continues (weak scaling).

Prints ONE JSON line on rank 0 with
  `roofline`      all conv MFMA launches of a step, timed with HIP event pairs on the launch stream;
  `cpu_baseline`  the oracle (a port of the reference path) timed on this host on a bounded sample;
  `parity`        loss / pose deviation of the TIMED dtype from the oracle on one step of the full workload
                  (identical batch and weights; the oracle is the checker, never the thing measured).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_IMAGE_TRAIN = 39.06   # 3 x 13.02 GFLOP (fwd + dgrad + wgrad), conv + linear
GFLOP_PER_IMAGE_EXECUTED = 38.65  # the stem's unused input gradient is not computed
PEAK_F16_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense fp16/bf16 MFMA
PEAK_F32_TFLOPS = 157.3
PROFILE_ROUND = "r02"


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _oracle_setup(windows, H, W):
    import torch
    import oracle
    torch.manual_seed(7)
    net = oracle.MapNet(oracle.PoseNet(oracle.resnet34(), droprate=0.0, pretrained=False))
    crit = oracle.MapNetCriterion(0.0, -3.0, 0.0, -3.0, True, True)
    opt = oracle.Optimizer([{"params": net.parameters()}, {"params": [crit.sax, crit.saq]},
                            {"params": [crit.srx, crit.srq]}], "adam", base_lr=1e-4, weight_decay=5e-4)
    x, t = oracle.make_batch("mapnet", windows, H, W, seed=7)
    net.train()
    return oracle, net, crit, opt, x, t


def cpu_baseline_and_parity(args, dev, binding=None):
    """The oracle leg (rank 0, N = 1 only).  (1) One oracle step of the FULL workload on the host: it is the checker
    for `parity` -- a fresh HIP model with the oracle's initial weights takes the same step on the same batch in the
    dtype that was timed -- and a single-step CPU timing.  (2) The bounded CPU timing sample SURVEY.md 8(d) asks for:
    5 windows, 3 warm-up + 10 timed steps, median."""
    import torch
    import geomapnet_amd as G
    n, H, W = args.windows, args.height, args.width
    out = {}
    # ---- (1) parity of the timed mode at the full workload
    oracle, onet, ocrit, oopt, x, t = _oracle_setup(n, H, W)
    kw = {} if binding is None else {"_binding": binding}
    net = G.MapNet(G.PoseNet(G.resnet34(**kw), droprate=0.0, pretrained=False, **kw))
    net.load_state_dict(onet.state_dict())
    if dev.type == "cuda":
        net.cuda()
    crit = G.MapNetCriterion(sax=0.0, saq=-3.0, srx=0.0, srq=-3.0, learn_beta=True, learn_gamma=True, **kw)
    if dev.type == "cuda":
        crit.cuda()
    opt = G.Optimizer([{"params": net.parameters()}, {"params": [crit.sax, crit.saq]}, {"params": [crit.srx, crit.srq]}],
                      "adam", base_lr=1e-4, weight_decay=5e-4)
    net.train()
    l, p = G.step_feedfwd(x.to(dev), net, dev.type == "cuda", t.to(dev), crit, opt, True)
    p = p.cpu()
    del net, crit, opt
    t0 = time.perf_counter()
    lo, po = oracle.step_feedfwd(x, onet, False, t, ocrit, oopt, True)
    full_s = time.perf_counter() - t0
    po = po.detach()
    out["parity"] = {
        "dtype": args.dtype, "checker": "oracle (CPU fp32 port of the reference path), same batch, same initial weights, step 1",
        "config": "%d windows x T=3 = %d images %dx%d" % (n, n * 3, H, W),
        "loss": round(float(l), 6), "loss_oracle": round(float(lo), 6),
        "loss_rel": float("%.3e" % (abs(l - lo) / max(1.0, abs(lo)))),
        "pose_abs_max": float("%.3e" % (p - po).abs().max().item()),
        "pose_abs_rms": float("%.3e" % (p - po).pow(2).mean().sqrt().item()),
        "pose_scale_max": float("%.3e" % po.abs().max().item()),
        "bar": "north star: 1e-4 on loss (read as relative to max(1,|loss|)), 1e-3 on pose (max abs)",
    }
    del onet, ocrit, oopt, x, t
    # ---- (2) bounded timing sample
    sw, warm, timed = 5, 3, 10
    if args.emu:
        sw, warm, timed = 1, 0, 1
    oracle, onet, ocrit, oopt, x, t = _oracle_setup(sw, H, W)
    for _ in range(warm):
        oracle.step_feedfwd(x, onet, False, t, ocrit, oopt, True)
    ts = []
    for _ in range(timed):
        t0 = time.perf_counter()
        oracle.step_feedfwd(x, onet, False, t, ocrit, oopt, True)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    out["cpu_baseline"] = {
        "value": round(sw * 3 / med, 2), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
        "cpu": cpu_model(),
        "sample": "oracle MapNet train step (fwd+loss+bwd+Adam), fp32, %d windows x T=3 = %d images %dx%d, "
                  "%d warm-up + %d timed steps, median %.3f s/step; one step of the full workload (%d images): %.1f s = "
                  "%.2f images/s" % (sw, sw * 3, H, W, warm, timed, med, n * 3, full_s, n * 3 / full_s)}
    return out


def self_launch(args):
    """python bench.py --gpus N (N > 1) outside any launcher: become the launcher."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--windows", type=int, default=64, help="windows per GPU per step")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=341)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle leg (cpu_baseline + parity)")
    ap.add_argument("--no-events", action="store_true", help="do not time conv launches with HIP events")
    ap.add_argument("--emu", action="store_true",
                    help="TEST ONLY: run the same code on the CPU SIMT-emulator build of the kernels over gloo "
                         "(tests/test_bench_launch.py); never a measurement")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ
    if not launched and args.gpus > 1:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))

    import torch
    import torch.distributed as dist
    binding = None
    if args.emu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import emu_lib
        binding = emu_lib.load()
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available() or torch.cuda.device_count() <= local:
            raise SystemExit("bench.py: rank %d needs GPU %d, %d visible; the HIP path has no CPU fallback"
                             % (rank, local, torch.cuda.device_count() if torch.cuda.is_available() else 0))
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.emu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import geomapnet_amd as G
    from geomapnet_amd.posenet import engine_of
    G.set_compute_dtype(args.dtype)
    kw = {} if binding is None else {"_binding": binding}
    torch.manual_seed(7)
    net = G.MapNet(G.PoseNet(G.resnet34(**kw), droprate=0.0, pretrained=False, **kw))
    crit = G.MapNetCriterion(sax=0.0, saq=-3.0, srx=0.0, srq=-3.0, learn_beta=True, learn_gamma=True, **kw)
    if dev.type == "cuda":
        net.cuda()
        crit.cuda()
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

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        sync()

    eng = engine_of(net)
    losses = []
    for _ in range(args.warmup):
        l, _ = G.step_feedfwd(images, net, dev.type == "cuda", targets, crit, opt, True)
        losses.append(l)
    use_events = not args.no_events and not args.emu

    import ctypes as C
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        l, _ = G.step_feedfwd(images, net, dev.type == "cuda", targets, crit, opt, True)
        losses.append(l)
    sync()
    barrier()
    elapsed = time.perf_counter() - t0
    ranks_seen = 1
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
        one = torch.ones(1, device=dev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)  # through the same RCCL communicator the gradient buckets use
        ranks_seen = int(one.item())
    plan = next(iter(eng.plans.values()))

    # Kernel-level roofline: the conv MFMA launches are timed with HIP event pairs on the launch stream over a second
    # run of the SAME K steps (event records serialise the two streams, so this run is not the one `value` comes from).
    conv_ms, conv_regions, eager_ms = 0.0, 0, None
    if use_events:  # every rank takes part: with world > 1 each step contains collectives
        eng.lib.check(eng.lib.set_profiling(plan["handle"], 1))
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            G.step_feedfwd(images, net, True, targets, crit, opt, True)
            ms, cnt = C.c_float(), C.c_int()
            eng.lib.check(eng.lib.last_kernel_ms(plan["handle"], 0, C.byref(ms), C.byref(cnt)))
            conv_ms += ms.value
            conv_regions += cnt.value
        sync()
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
            for rnd in (PROFILE_ROUND, "r01"):
                try:
                    with open(os.path.join(ROOT, "profiles", rnd, "pmc_conv_traffic.json")) as f:
                        traffic = json.load(f)["hbm_bytes_per_step"] if (args.dtype == "fp16" and n == 64) else None
                    break
                except Exception:
                    pass
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": traffic,
                    "kernel": "all conv MFMA kernels of a step (igemm + conv_halo + wgrad): %d convolution operators per step, "
                              "each timed as one region (a stride-2 data gradient is 2-4 launches)" % (conv_regions // args.steps),
                    "conv_ms_per_step": round(conv_ms_step, 3), "eager_profiled_ms_per_step": round(eager_ms, 3),
                    "flops_per_step_G": round(GFLOP_PER_IMAGE_TRAIN * n * T, 1),
                    "whole_step_frac": round(GFLOP_PER_IMAGE_TRAIN * n * T / ms_per_step / peak, 4)}
        out = {"metric": "images/sec MapNet ResNet-34 256x341 T=3 train step", "value": round(value, 2), "unit": "images/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f16" if args.dtype == "fp16" else "f32", "data": "synthetic",
               "config": {"workload": "BASELINE configs[2]: MapNet ResNet-34, %d windows x T=3 = %d images/GPU/step, %dx%d, "
                                      "MapNetCriterion learned beta/gamma, Adam, random-init weights" % (n, n * T, H, W),
                          "global_windows": n * world, "parallelism": "dp%d" % world, "n_ranks_seen": ranks_seen,
                          "launcher": "torch.distributed.run" if launched else "none (single process)",
                          "loss_first": round(losses[0], 4), "loss_last": round(losses[-1], 4),
                          "gflop_per_image": GFLOP_PER_IMAGE_TRAIN, "gflop_per_image_executed": GFLOP_PER_IMAGE_EXECUTED},
               "roofline": roof}
        if args.emu:
            out["data"] = "synthetic (CPU emulator dry-run: NOT a measurement)"
        if world == 1 and not args.no_cpu_baseline:
            del net, crit, opt, images
            try:
                out.update(cpu_baseline_and_parity(args, dev, binding))
            except Exception as e:  # the oracle leg must never hide the GPU number
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
