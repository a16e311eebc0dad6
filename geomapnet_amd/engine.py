"""Host-side owner of the device arenas and of the mn_handle plans (plumbing, not math).

PyTorch tensors are used purely as containers for HBM (parameter arena, optimiser state,
BatchNorm buffers, per-plan work arena) and as the source of the current HIP stream; all
arithmetic happens in libmapnet_hip.so.
"""
import ctypes as C
import os

import torch

from . import _binding
from ._binding import Config, MapNetHipError, ptr

MODE_POSENET, MODE_MAPNET, MODE_ONLINE, MODE_GPS = 0, 1, 2, 3
# "fp32x3": fp32 tensors exactly as "fp32", every convolution contracted on the f16 / bf16 matrix pipe with operands
# split into hi + lo halves in registers (three MFMAs per product; include/mapnet_hip.h MN_DTYPE_F32X3)
# "fp16x2": fp32-class values stored as fp16 PAIRS (hi + lo halves, 4 bytes per element) wherever a convolution consumes them,
# split once by the producing kernel; the convolutions run three fp16 MFMAs per product on DMA-fed operands; conv outputs,
# data gradients, BatchNorm, head, criterion and optimiser in fp32 (include/mapnet_hip.h MN_DTYPE_F16X2).  Gradients live in
# fp16 pairs, so the mode uses the fp16 mode's loss scale and overflow guard.
# "fp16x2m": the forward pass of "fp16x2" bit for bit (loss and poses are that mode's) and a backward pass on the "fp16" kernels --
# one MFMA per product on single fp16 operands, gates and BatchNorm statistics from the exact forward values
# (include/mapnet_hip.h MN_DTYPE_F16X2M; what it costs the gradients: tools/mixed_budget.py)
# "fp16x2q" (experimental): fp16x2m whose forward convolutions take both cross terms from fp8 copies on the block-scaled MFMA
# (MN_DTYPE_F16X2Q): 2 instead of 3 MFMA-equivalents per forward product, poses ~3e-4 from the fp32 reference instead of 1.6e-5
DTYPES = {"fp32": 0, "fp16": 1, "fp32x3": 2, "fp16x2": 3, "fp16x2m": 4, "fp16x2q": 5}
SCALED_DTYPES = ("fp16", "fp16x2", "fp16x2m", "fp16x2q")  # modes whose gradients pass through fp16 halves

# The library default is the mode inside the north-star tolerance (loss 1e-4 / poses 1e-3 against the fp32 reference): the same
# default as scripts/train.py, scripts/eval.py and bench.py (round-5 ADVICE: API users and script users must not get different
# numerics).  set_compute_dtype("fp16") buys ~1.5x the throughput at poses 1.3e-2 from the reference.
_default_dtype = "fp16x2m"
_default_loss_scale = 1024.0

# Where a plan's work arena comes from: None = torch.empty (the caching allocator), or a callable (nbytes, device) -> uint8
# tensor of at least nbytes (tools/arena_probe.py places arenas at chosen offsets of one pooled allocation with it).
work_allocator = None
_step_streams = {}  # device index -> the process-wide step stream (Engine.step_stream)


def set_compute_dtype(name, loss_scale=None):
    """'fp16x2m' (default: fp16-pair conv operands in the forward pass -- loss and poses inside the north-star tolerance -- and the
    fp16 mode's one-MFMA backward pass), 'fp16' (fp16 tensors, fp32 accumulate: ~1.5x faster, outside the tolerance), 'fp16x2'
    (fp16-pair conv operands forward and backward, fp32 everything else),
    'fp32x3' (fp32 tensors, operands split inside the convolution kernels) or
    'fp32' (fp32 tensors on v_mfma_f32_32x32x2_f32, an exact fp32 FMA chain)."""
    global _default_dtype, _default_loss_scale
    if name not in DTYPES:
        raise ValueError(name)
    _default_dtype = name
    if loss_scale is not None:
        _default_loss_scale = float(loss_scale)


def get_compute_dtype():
    return _default_dtype


def _stream(t):
    if t.is_cuda:
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return None


class _StepLoss:
    """What `Engine.train_step` returns for the loss: `.item()` is the reference's blocking read-back
    (common/train.py:361) but only waits for the criterion, not for the rest of the step."""

    def __init__(self, engine, plan):
        self._e, self._p, self._v = engine, plan, None

    def item(self):
        if self._v is None:
            self._v = self._e._loss_value(self._p)
        return self._v

    def __float__(self):
        return self.item()


class Engine:
    """One per PoseNet: parameter/optimiser/buffer arenas + a cache of plans keyed by shape."""

    def __init__(self, feat_dim, binding=None, filter_nans=False):
        self.lib = binding if binding is not None else _binding.hip()
        self.feat_dim = feat_dim
        self.filter_nans = bool(filter_nans)
        self.entries = self.lib.entries(feat_dim)
        self.n_params = int(self.lib.model_param_floats(feat_dim))
        self.n_model = self.n_params - 4
        self.params = torch.zeros(self.n_params, dtype=torch.float32)
        self.buffers = torch.zeros(int(self.lib.model_buffer_bytes(feat_dim)), dtype=torch.uint8)
        self.opt_state = None  # [3 * n_params] grads | exp_avg | exp_avg_sq, created on first train step
        self.plans = {}
        self.version = 0  # bumped whenever parameters change outside a plan
        # Optimiser steps APPLIED (Adam's `step`).  The truth lives in the device counter of the plan that stepped last
        # (`_step_owner`; an fp16 step skipped on overflow does not advance it); `step_count` is the host's view: exact
        # after every read-back (effective_step, plan switch, drop_plans), otherwise it counts attempts.
        self.step_count = 0
        self._step_owner = None
        self.dtype = None  # None -> module default at plan creation
        self.loss_scale = None
        self.eps_mode = 0
        self.input_u8 = None  # (mean[3], std[3]) when images arrive as uint8 NHWC and are normalised on the device
        self.dropout = (0.0, 0)  # (p, Philox seed) of the device dropout on the feature vector (mn_set_dropout); p = 0: identity

    # -- arenas ---------------------------------------------------------------------------------
    @property
    def device(self):
        return self.params.device

    def move(self, fn):
        """apply a tensor -> tensor function (e.g. .cuda()) to every arena; plans are dropped."""
        new = fn(self.params)
        if new.dtype != torch.float32:
            raise MapNetHipError("the parameter arena is fp32; use set_compute_dtype() to choose the compute type")
        self.drop_plans()
        self.params = new
        self.buffers = self.buffers.to(new.device)
        if self.opt_state is not None:
            self.opt_state = self.opt_state.to(new.device)

    def drop_plans(self):
        self._read_step()
        self._step_owner = None
        for p in self.plans.values():
            self.lib.destroy(p["handle"])
        self.plans = {}

    def _read_step(self):
        """host view := the owning plan's device counter (waits for the device)"""
        if self._step_owner is not None:
            v = int(self.lib.get_step_count(self._step_owner["handle"]))
            if v >= 0:
                self.step_count = v
        return self.step_count

    def set_step(self, step):
        """a loaded optimiser state: `step` becomes the count every plan continues from"""
        self.step_count = int(step)
        self._step_owner = None

    def _own_step(self, p):
        """plan `p` is about to run: its device counter continues from the applied count, wherever that lives"""
        if self._step_owner is p:
            return
        self._read_step()
        self.lib.check(self.lib.set_step_count(p["handle"], self.step_count))
        if self.dropout[0] > 0.0:  # the mask sequence continues from the model's step count, not from this plan's call 0
            self.lib.check(self.lib.set_dropout_calls(p["handle"], self.step_count & 0xffffffff))
        self._step_owner = p

    def __del__(self):
        try:
            self.drop_plans()
        except Exception:
            pass

    def _check_device(self):
        if self.lib.backend_name == "hip" and not self.params.is_cuda:
            raise MapNetHipError("the MapNet HIP path needs the model on a GPU (model.cuda()); there is no CPU fallback")

    def ensure_opt_state(self):
        if self.opt_state is None:
            self.opt_state = torch.zeros(3 * self.n_params, dtype=torch.float32, device=self.device)
        return self.opt_state

    def grads(self):
        return self.opt_state[: self.n_params]

    def crit_slice(self):
        return self.params[self.n_model:]

    # -- plans ------------------------------------------------------------------------------------
    def plan(self, mode, windows, T, H, W):
        self._check_device()
        dtype = self.dtype or _default_dtype
        scale = self.loss_scale if self.loss_scale is not None else (_default_loss_scale if dtype in SCALED_DTYPES else 1.0)
        key = (mode, windows, T, H, W, dtype, scale, self.eps_mode)
        p = self.plans.get(key)
        if p is None:
            self.ensure_opt_state()
            cfg = Config(mode=mode, dtype=DTYPES[dtype], windows=windows, T=T, H=H, W=W, feat_dim=self.feat_dim,
                         filter_nans=int(self.filter_nans), loss_scale=scale, eps_mode=self.eps_mode)
            nbytes = int(self.lib.plan_bytes(C.byref(cfg)))
            if nbytes < 0:
                raise MapNetHipError(self.lib.last_error().decode())
            work = (work_allocator(nbytes, self.device) if work_allocator is not None
                    else torch.empty(nbytes, dtype=torch.uint8, device=self.device))
            h = self.lib.create(C.byref(cfg), ptr(self.params), ptr(self.opt_state), ptr(self.buffers), ptr(work),
                                _stream(self.params))
            if not h:
                raise MapNetHipError(self.lib.last_error().decode())
            frames = 1 if mode == MODE_POSENET else (T if mode == MODE_MAPNET else 2 * T)
            p = {"handle": C.c_void_p(h), "work": work, "version": self.version, "cfg": cfg, "images": windows * frames,
                 "loss": torch.zeros(1, dtype=torch.float32, device=self.device), "dtype": dtype}
            self.plans[key] = p
        if p["version"] != self.version:
            self.lib.check(self.lib.params_changed(p["handle"]))
            p["version"] = self.version
        if p.get("input_u8") != self.input_u8:
            if self.input_u8 is None:
                self.lib.check(self.lib.set_input_u8(p["handle"], 0, None, None))
            else:
                mean, std = ((C.c_float * 3)(*[float(v) for v in vals]) for vals in self.input_u8)
                self.lib.check(self.lib.set_input_u8(p["handle"], 1, mean, std))
            p["input_u8"] = self.input_u8
        if p.get("dropout", (0.0, 0)) != self.dropout:
            self.lib.check(self.lib.set_dropout(p["handle"], C.c_float(self.dropout[0]), C.c_uint64(self.dropout[1])))
            p["dropout"] = self.dropout
            if self.dropout[0] > 0.0:
                # mn_set_dropout restarts the mask sequence at 0: continue from the model's count of APPLIED steps, which lives on
                # the device of the plan that stepped last (round-5 ADVICE: the host's view is stale while that plan owns the
                # counter, and _own_step returns early for the owner -- the sequence would replay its first masks)
                self._read_step()
                self.lib.check(self.lib.set_dropout_calls(p["handle"], self.step_count & 0xffffffff))
        self._own_step(p)
        return p

    def params_touched(self):
        self.version += 1

    # -- fp16 loss scaling / overflow bookkeeping (mn_get_loss_scale) ------------------------------------------
    def loss_scale_state(self, sync=True):
        """-> (scale the next step will use, training steps skipped because their gradients overflowed), summed over
        this engine's plans; sync=True waits for the device first so the count is exact."""
        if sync and self.params.is_cuda:
            torch.cuda.synchronize(self.device)
        scale, skipped = None, 0
        for p in self.plans.values():
            sc, sk = C.c_float(), C.c_int64()
            self.lib.check(self.lib.get_loss_scale(p["handle"], C.byref(sc), C.byref(sk)))
            scale = sc.value if scale is None else min(scale, sc.value)
            skipped += sk.value
        return scale, skipped

    def effective_step(self):
        """optimiser steps actually applied (Adam's `step`; fp16 steps skipped on overflow are not steps): read back from
        the device counter of the plan that stepped last"""
        return self._read_step()

    def check_overflow_progress(self, p, limit=8):
        """fp16: steps that overflowed although the loss scale already was 1 cannot be fixed by scaling (a NaN input, a
        forward pass that overflowed); the device skips them forever.  Raise instead of printing losses that train nothing."""
        stuck = int(self.lib.stuck_overflow_steps(p["handle"]))
        if stuck >= limit:
            raise MapNetHipError("%d consecutive training steps were skipped with non-finite gradients at loss scale 1: the inputs or "
                                 "the forward pass are not finite (fp16 range?); no parameter has been updated since" % stuck)

    def debug_tensor(self, plan, name):
        """a named activation / gradient of the plan's work arena as a tensor VIEW (tests, tools/layer_error.py)"""
        ptr_, n, dt = C.c_void_p(), C.c_int64(), C.c_int32()
        self.lib.check(self.lib.debug_tensor(plan["handle"], name.encode(), C.byref(ptr_), C.byref(n), C.byref(dt)))
        work = plan["work"]
        off = ptr_.value - work.data_ptr()
        if dt.value not in (0, 1, 3, 5):
            raise MapNetHipError("debug_tensor: unknown dtype code %d for %r" % (dt.value, name))
        es = 2 if dt.value == 1 else 4
        assert 0 <= off and off + n.value * es <= work.numel()
        raw = work[off: off + n.value * es]
        if dt.value == 5:  # h2q: per 32 channels 32 fp16 hi halves, 32 fp8 lo values (x 2^-9), 32 fp8 copies of hi -> hi + lo, a COPY
            g = raw.view(-1, 128)
            hi = g[:, :64].contiguous().view(torch.float16).float()
            lo = g[:, 64:96].contiguous().view(torch.float8_e4m3fn).float() * 2.0 ** -9
            return (hi + lo).reshape(-1)
        if dt.value == 3:  # h2 (fp16 pairs): per 32 channels 32 hi halves then 32 lo halves -> fp32 values hi + lo, a COPY
            h = raw.view(torch.float16).view(-1, 2, 32).float()
            return (h[:, 0, :] + h[:, 1, :]).reshape(-1)
        return raw.view(torch.float16 if dt.value == 1 else torch.float32)

    # -- calls --------------------------------------------------------------------------------------
    def set_dropout(self, p, seed=0):
        """device dropout on the feature vector in training forward passes (include/mapnet_hip.h mn_set_dropout); p = 0 disables"""
        self.dropout = (float(p), int(seed))

    def dropout_mask(self, plan):
        """[images, feat_dim] mask (0 or 1/(1-p)) the last training forward pass of `plan` applied (parity tooling)"""
        ptr_, n, dt = C.c_void_p(), C.c_int64(), C.c_int32()
        self.lib.check(self.lib.debug_tensor(plan["handle"], b"dropmask", C.byref(ptr_), C.byref(n), C.byref(dt)))
        if self.params.is_cuda:
            torch.cuda.synchronize(self.device)
        buf = (C.c_float * n.value).from_address(ptr_.value) if not self.params.is_cuda else None
        if buf is not None:
            return torch.frombuffer(buf, dtype=torch.float32).clone().view(plan["images"], self.feat_dim)
        out = torch.empty(n.value, dtype=torch.float32, device=self.device)
        off = ptr_.value - plan["work"].data_ptr()
        out.copy_(plan["work"][off:off + 4 * n.value].view(torch.float32))
        return out.view(plan["images"], self.feat_dim)

    def set_input_u8(self, mean=None, std=None):
        """images become uint8 [.., H, W, 3]; (x/255 - mean)/std runs on the device.  mean=None: back to fp32 NCHW."""
        self.input_u8 = None if mean is None else (tuple(float(v) for v in mean), tuple(float(v) for v in std))

    def image_dims(self, images):
        """-> (H, W) of one frame, for either input format; validates the dtype against the configured format"""
        if self.input_u8 is not None:
            if images.dtype != torch.uint8 or images.shape[-1] != 3:
                raise MapNetHipError("uint8 input mode expects uint8 images [..., H, W, 3]")
            return int(images.shape[-3]), int(images.shape[-2])
        if images.dtype != torch.float32:
            raise MapNetHipError("expected fp32 images [..., 3, H, W] (or call set_input_u8 for uint8 NHWC input)")
        return int(images.shape[-2]), int(images.shape[-1])

    def forward(self, images, training):
        """images: fp32 [B,3,H,W] (or uint8 [B,H,W,3]) contiguous on the engine's device -> poses [B,6]."""
        B = images.shape[0]
        H, W = self.image_dims(images)
        p = self.plan(MODE_POSENET, B, 1, H, W)
        out = torch.empty(B, 6, dtype=torch.float32, device=self.device)
        self.lib.check(self.lib.forward(p["handle"], ptr(images), ptr(out), int(bool(training)), _stream(images)))
        return out

    def configure_step(self, p, lr, weight_decay, betas, eps, max_grad_norm, learn_beta, learn_gamma, method=(0, 0)):
        h = p["handle"]
        self.lib.check(self.lib.set_optim(h, lr, weight_decay, betas[0], betas[1], eps, max_grad_norm))
        if p.get("method", (0, 0)) != tuple(method):
            self.lib.check(self.lib.set_optim_method(h, int(method[0]), int(method[1])))
            p["method"] = tuple(method)
        flags = (bool(learn_beta), bool(learn_gamma))
        if p.get("flags") != flags:
            self.lib.check(self.lib.set_learn_flags(h, int(flags[0]), int(flags[1])))
            p["flags"] = flags

    # -- training steps run on a stream of the engine's own, ordered after the caller's current stream and joined back
    #    into it: the library forks weight-gradient work onto a second (non-blocking) stream, which the legacy default
    #    stream's implicit synchronisation would serialise
    def step_stream(self):
        """ONE step stream per device for the whole process, created before the library's weight-gradient side stream.
        HIP streams are multiplexed onto a small number of hardware queues (GPU_MAX_HW_QUEUES, 4 by default) in creation order, and
        two streams that land on the same queue execute in order: round 5's "every fourth model of a process is 7-11 % slow" was a
        new torch stream per Engine walking round that ring until the step stream shared a queue with the library's side stream --
        the step lost the overlap of its weight-gradient launches, exactly the ~1.3 ms the second stream buys
        (tools/arena_probe.py, profiles/r06/c1_fourth_model_root_cause.txt: the slow model follows the stream count, not the
        arena's address, offset or physical backing)."""
        if not self.params.is_cuda:
            return None
        key = self.device.index if self.device.index is not None else torch.cuda.current_device()
        s = _step_streams.get(key)
        if s is None:
            s = _step_streams[key] = torch.cuda.Stream(device=self.device)
        return s

    def train_step(self, p, images, targets):
        if "poses" not in p:
            p["poses"] = torch.empty(p["images"], 6, dtype=torch.float32, device=self.device)
            if self.params.is_cuda and os.environ.get("MN_EARLY_LOSS", "1") != "0":
                p["loss_host"] = torch.zeros(1, dtype=torch.float32).pin_memory()
                self.lib.check(self.lib.set_loss_host(p["handle"], ptr(p["loss_host"])))
        side = self.step_stream()
        if side is None:
            self.lib.check(self.lib.train_step(p["handle"], ptr(images), ptr(targets), ptr(p["loss"]), ptr(p["poses"]), None))
            self._stepped(p)
            return p["loss"], p["poses"].clone()
        cur = torch.cuda.current_stream(self.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self.lib.check(self.lib.train_step(p["handle"], ptr(images), ptr(targets), ptr(p["loss"]), ptr(p["poses"]),
                                               C.c_void_p(side.cuda_stream)))
            poses = p["poses"].clone()
        cur.wait_stream(side)
        self._stepped(p)
        return _StepLoss(self, p), poses

    def _loss_value(self, p):
        """this step's loss as a Python float: the early host copy when the library posted one, else the device scalar"""
        if "loss_host" in p and self.lib.wait_loss(p["handle"]) == 0:
            return float(p["loss_host"][0])
        return p["loss"].item()

    def _stepped(self, p):
        self.step_count += 1
        if p.get("dtype") in SCALED_DTYPES and self.step_count % 16 == 0:
            self.check_overflow_progress(p)
        self.version += 1
        p["version"] = self.version  # this plan repacks by itself after its own optimiser step
