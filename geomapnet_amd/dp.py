"""Data parallelism over the GPUs of one node: one process per GPU, torch.distributed
(backend 'nccl' = RCCL over xGMI) as the transport, gradients all-reduced per bucket while the
rest of the backward pass is still running.

The reference has no multi-GPU path; this is new functionality.  Windows are
independent in the network and in every loss term, so ranks shard WINDOWS (never frames of a
window); each rank keeps a full replica and normalises BatchNorm over its local batch, exactly
as N independent reference processes would.  The gradient of the global mean loss is the mean of
the rank gradients: buckets are summed by all-reduce and the 1/world factor is folded into the
fused Adam kernel (`grad_mul`).

Bucket b = parameters of stage b (3 = layer4+fc+heads+criterion scalars, 2 = layer3, 1 = layer2,
0 = stem+layer1), contiguous ranges of the flat gradient arena, issued in backward order.
xGMI is point-to-point (7 links x ~153 GB/s): the 57 MB stage-3 bucket is in flight while ~80 % of
the backward FLOPs (layers 3..1) are still to run.
"""
import torch
import torch.distributed as dist

from ._binding import ptr
from .engine import _stream


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


# Measurement hook (bench.py): with profiling on, every staged step brackets its wait for the gradient all-reduces with two
# timed events on the compute stream.  Their distance is the EXPOSED communication time of the step: how long the optimiser
# was held back by collectives that backward did not hide (0 when every bucket had landed before the last stage ended).
# It also records a per-bucket timeline (bucket_timeline_ms): when each backward stage had finished on the compute stream
# (= the moment its bucket's all-reduce could start) and when the compute stream got past the wait for that bucket, both
# relative to the start of the backward pass -- a bucket whose `passed` is later than the last stage's `ready` was exposed.
_profile = {"on": False, "pairs": [], "timelines": []}


def set_profiling(on):
    _profile["on"] = bool(on)
    _profile["pairs"] = []
    _profile["timelines"] = []


def bucket_timeline_ms():
    """-> per profiled step {stage: (ready_ms, passed_ms)} relative to the start of the backward pass; synchronises"""
    out = []
    for t0, ready, passed in _profile["timelines"]:
        passed[-1][1].synchronize()
        out.append({st: (round(t0.elapsed_time(ev), 3), round(t0.elapsed_time(dict(passed)[st]), 3)) for st, ev in ready})
    _profile["timelines"] = []
    return out


def exposed_comm_ms():
    """-> list of exposed-communication times (ms), one per profiled step; synchronises the recorded events"""
    out = []
    for a, b in _profile["pairs"]:
        b.synchronize()
        out.append(a.elapsed_time(b))
    _profile["pairs"] = []
    return out


def train_step(engine, plan, images, targets):
    import ctypes as C
    import contextlib
    lib, h = engine.lib, plan["handle"]
    if "poses" not in plan:
        plan["poses"] = torch.empty(plan["images"], 6, dtype=torch.float32, device=engine.device)
    side = engine.step_stream()
    if side is not None:  # the step runs on the engine's own stream (ordered after the caller's)
        cur = torch.cuda.current_stream(engine.device)
        side.wait_stream(cur)
        ctx = torch.cuda.stream(side)
        s = C.c_void_p(side.cuda_stream)
    else:
        ctx = contextlib.nullcontext()
        s = None
    with ctx:
        loss, poses = _staged_step(engine, plan, images, targets, lib, h, s)
    if side is not None:
        cur.wait_stream(side)
    return loss, poses


def _staged_step(engine, plan, images, targets, lib, h, s):
    import ctypes as C
    poses = plan["poses"]
    lib.check(lib.train_forward_loss(h, ptr(images), ptr(targets), ptr(plan["loss"]), ptr(poses), s))
    grads = engine.grads()
    works = []
    multi = world_size() > 1 or (dist.is_available() and dist.is_initialized())
    timed = _profile["on"] and s is not None
    if timed:
        t0 = torch.cuda.Event(enable_timing=True)
        t0.record()
        ready, passed = [], []
    for stage in (3, 2, 1, 0):
        lib.check(lib.train_backward_stage(h, stage, s))
        if timed:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ready.append((stage, e))
        off, cnt = C.c_int64(), C.c_int64()
        lib.check(lib.grad_bucket(h, stage, C.byref(off), C.byref(cnt)))
        bucket = grads[off.value: off.value + cnt.value]
        # async: RCCL runs on its own stream, ordered after the kernels enqueued so far
        if multi:
            works.append(dist.all_reduce(bucket, op=dist.ReduceOp.SUM, async_op=True))
    if timed:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    for i, w in enumerate(works):
        w.wait()  # stream-level wait on CUDA/HIP; blocking on gloo
        if timed:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            passed.append((ready[i][0], e))
    if timed:
        ev[1].record()
        _profile["pairs"].append(ev)
        if passed:
            _profile["timelines"].append((t0, ready, passed))
    lib.check(lib.optim_step(h, 1.0 / world_size(), s))
    engine._stepped(plan)
    # reported loss = mean of the rank losses (one scalar all-reduce)
    loss = plan["loss"].clone()
    if multi:
        dist.all_reduce(loss, op=dist.ReduceOp.SUM)
        loss /= world_size()
    return loss, poses.clone()
