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
import os

import torch
import torch.distributed as dist

from ._binding import ptr
from .engine import _stream

# ---- schedule knobs (read per step, so tools can flip them inside one process) ----------------------------------------------
# MN_DP_DEFER: when the buckets' all-reduces are ISSUED.  0 (default): each right after its backward stage (maximum overlap);
#   1: bucket 3 (57 MB, layer4 + head) right after stage 3, buckets 2..0 after backward_stage(0) -- they would otherwise sit on the
#   CUs beside layer3 / layer2 / layer1 launches that are sized as exactly one round of the chip; 2: every bucket after the last
#   stage (no overlap, no contention).  The rehearsal table (profiles/r06/rccl_rehearsal.txt) is what the default was chosen from.
# MN_DP_STANDIN="c[,threads[,busbw_GBps[,latency_us[,lds_kb]]]]": ONE-GPU rehearsal of an 8-GPU run -- instead of an all-reduce (a 1-rank
#   all-reduce launches nothing) each bucket launches the library's occupancy stand-in (mn_op_occupy) on a communication stream:
#   c workgroups of `threads` threads resident for the time a ring all-reduce of the bucket over 8 GPUs would take at `busbw`
#   (default 200 GB/s bus bandwidth + 40 us latency), streaming the bucket's reduce traffic through HBM meanwhile; lds_kb (0, 32,
#   64): LDS each stand-in workgroup holds -- with tens of KB a CU cannot host a 150 KB convolution workgroup beside it.
# MN_DP_GRAD_DTYPE=fp32|bf16|auto (default fp32): element type the buckets travel in.  bf16 (auto = bf16 for the plain fp16 mode, fp32
#   otherwise): the library rounds a bucket to bf16 (mn_grad_bucket_pack_bf16), the all-reduce runs on the 2-byte values, and
#   mn_grad_bucket_unpack_bf16 widens them back in place before the optimiser -- half the bytes on xGMI and half the time the
#   collective's workgroups share the CUs.  Costs one rounding of every rank's gradient to 8 bits and bf16 partial sums inside the
#   ring: far below the fp16 mode's own gradient deviation (16 %), far ABOVE fp16x2m's backward rounding budget (8e-4), which is why
#   the parity mode keeps fp32 transport unless asked.  (fp16 is not an option: weight gradients times the loss scale overflow it.)
RING_WORLD = 8  # the world the stand-in's duration model assumes


def _defer_mode():
    return int(os.environ.get("MN_DP_DEFER", "0"))


def _standin():
    v = os.environ.get("MN_DP_STANDIN", "")
    if not v:
        return None
    f = v.split(",")
    return {"c": int(f[0]), "threads": int(f[1]) if len(f) > 1 else 256, "busbw": float(f[2]) if len(f) > 2 else 200.0,
            "lat_us": float(f[3]) if len(f) > 3 else 40.0, "lds_kb": int(f[4]) if len(f) > 4 else 0}


def _grad_transport(plan):
    v = os.environ.get("MN_DP_GRAD_DTYPE", "fp32")
    if v == "auto":
        v = "bf16" if plan.get("dtype") == "fp16" else "fp32"
    if v not in ("fp32", "bf16"):
        raise ValueError("MN_DP_GRAD_DTYPE must be fp32, bf16 or auto")
    return v


def ring_allreduce_us(nbytes, world=RING_WORLD, busbw_GBps=200.0, lat_us=40.0):
    """duration model of a ring all-reduce: every GPU moves 2 (world-1)/world of the bucket over its links"""
    return lat_us + nbytes * 2.0 * (world - 1) / world / (busbw_GBps * 1e3)


def rccl_env(defaults=None):
    """RCCL channel budget for the data-parallel step, set (setdefault) BEFORE the process group is created: every RCCL channel is
    one resident workgroup per collective, and the backward convolutions are launches of one workgroup per CU sized as one round
    of the chip -- each channel beyond the 20-odd CUs those launches leave idle pushes a convolution tile into a second round
    (the stand-in rehearsal, profiles/r06/rccl_rehearsal.txt).  The buckets are 0.9-57 MB under >= 2 ms of backward each: they
    need residency discipline, not peak bus bandwidth."""
    # (GPU_MAX_HW_QUEUES is read when the HIP runtime loads, i.e. at `import torch`: a launcher has to export it -- bench.py does; two
    #  streams that share one of HIP's 4 default hardware queues run in order, and a data-parallel rank has five streams)
    env = {"NCCL_MIN_NCHANNELS": "4", "NCCL_MAX_NCHANNELS": "8"}
    if defaults:
        env.update(defaults)
    for k, v in env.items():
        os.environ.setdefault(k, v)
    return {k: os.environ[k] for k in env}


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


class _StandinWork:
    """what `dist.all_reduce(async_op=True)` returns, for the one-GPU stand-in: wait() = the compute stream waits for the event"""

    def __init__(self, ev):
        self._ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self._ev)


_standin_state = {}


def _standin_launch(engine, lib, bucket, cfg):
    """the stand-in kernel on a communication stream of its own, ordered after the compute stream as RCCL's stream is"""
    import ctypes as C
    dev = engine.device
    st = _standin_state.get(dev)
    if st is None:
        st = _standin_state[dev] = {"stream": torch.cuda.Stream(device=dev), "src": None, "dst": None}
    nbytes = bucket.numel() * bucket.element_size()
    if st["src"] is None or st["src"].numel() < nbytes:  # (scratch: the stand-in must not touch the gradients)
        st["src"] = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        st["dst"] = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    us = ring_allreduce_us(nbytes, RING_WORLD, cfg["busbw"], cfg["lat_us"])
    comm = st["stream"]
    comm.wait_stream(torch.cuda.current_stream(dev))
    # local HBM traffic of a ring step: every byte of the bucket is read ~2x and written ~2x on each GPU (reduce-scatter + all-gather)
    lib.check(lib.op_occupy(cfg["c"], cfg["threads"], C.c_float(us), ptr(st["src"]), ptr(st["dst"]), C.c_int64((nbytes // 16) * 16),
                            cfg["lds_kb"], C.c_void_p(comm.cuda_stream)))
    ev = torch.cuda.Event()
    ev.record(comm)
    return _StandinWork(ev)


def _staged_step(engine, plan, images, targets, lib, h, s):
    import ctypes as C
    poses = plan["poses"]
    lib.check(lib.train_forward_loss(h, ptr(images), ptr(targets), ptr(plan["loss"]), ptr(poses), s))
    grads = engine.grads()
    works = []
    multi = world_size() > 1 or (dist.is_available() and dist.is_initialized())
    standin = _standin() if (s is not None and not multi) else None
    defer = _defer_mode()
    deferred = []
    timed = _profile["on"] and s is not None
    if timed:
        t0 = torch.cuda.Event(enable_timing=True)
        t0.record()
        ready, passed = [], []

    half = _grad_transport(plan) == "bf16" and (multi or standin is not None)
    unpack = []  # stages whose halves must be widened back into the arena once their all-reduce has landed

    def issue(bucket, stage):
        # async: RCCL runs on its own stream, ordered after the kernels enqueued so far
        if half:
            bufs = plan.setdefault("dp_bf16", {})
            if stage not in bufs:
                bufs[stage] = torch.empty(bucket.numel(), dtype=torch.bfloat16, device=bucket.device)
            lib.check(lib.grad_bucket_pack_bf16(h, stage, ptr(bufs[stage]), s))
            bucket = bufs[stage]
            unpack.append(stage)
        if multi:
            works.append(dist.all_reduce(bucket, op=dist.ReduceOp.SUM, async_op=True))
        elif standin is not None:
            works.append(_standin_launch(engine, lib, bucket, standin))

    for stage in (3, 2, 1, 0):
        lib.check(lib.train_backward_stage(h, stage, s))
        if timed:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ready.append((stage, e))
        off, cnt = C.c_int64(), C.c_int64()
        lib.check(lib.grad_bucket(h, stage, C.byref(off), C.byref(cnt)))
        bucket = grads[off.value: off.value + cnt.value]
        if defer == 2 or (defer == 1 and stage != 3):
            deferred.append((bucket, stage))  # issued after the last backward stage, in backward order
        else:
            issue(bucket, stage)
    for bucket, stage in deferred:
        issue(bucket, stage)
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
    for stage in unpack:  # (every all-reduce has been waited for on this stream)
        lib.check(lib.grad_bucket_unpack_bf16(h, stage, ptr(plan["dp_bf16"][stage]), s))
    lib.check(lib.optim_step(h, 1.0 / world_size(), s))
    engine._stepped(plan)
    # reported loss = mean of the rank losses (one scalar all-reduce)
    loss = plan["loss"].clone()
    if multi:
        dist.all_reduce(loss, op=dist.ReduceOp.SUM)
        loss /= world_size()
    return loss, poses.clone()
