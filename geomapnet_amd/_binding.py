"""ctypes binding of the C ABI declared in include/mapnet_hip.h.

`hip()` loads geomapnet_amd/libmapnet_hip.so -- the gfx950 build -- and is the only loader the
product uses.  There is no CPU fallback: if the library is missing or no GPU is visible the
product raises.  (`Binding(cdll)` can wrap any library exporting the same ABI; the test-suite
uses that to drive its SIMT-emulator build of the same kernel sources.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmapnet_hip.so")

c_void = C.c_void_p
c_f = C.c_float
c_i = C.c_int
c_i64 = C.c_int64


class Entry(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("offset", c_i64), ("numel", c_i64), ("ndim", C.c_int32),
                ("shape", C.c_int32 * 4), ("is_buffer", C.c_int32), ("is_int64", C.c_int32), ("ohwi", C.c_int32),
                ("stage", C.c_int32)]


class Config(C.Structure):
    _fields_ = [("mode", C.c_int32), ("dtype", C.c_int32), ("windows", C.c_int32), ("T", C.c_int32), ("H", C.c_int32),
                ("W", C.c_int32), ("feat_dim", C.c_int32), ("filter_nans", C.c_int32), ("loss_scale", c_f),
                ("eps_mode", C.c_int32)]


class GatherGeom(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "Hi", "Wi", "C", "P", "Q", "R", "S", "mul_p", "mul_q", "rsign", "ssign",
                                         "off_h", "off_w", "div", "M", "N", "K")]


_PROTOS = {
    "mn_last_error": (C.c_char_p, []),
    "mn_backend": (C.c_char_p, []),
    "mn_model_entries": (c_i, [c_i]),
    "mn_model_entry": (c_i, [c_i, c_i, C.POINTER(Entry)]),
    "mn_model_param_floats": (c_i64, [c_i]),
    "mn_model_buffer_bytes": (c_i64, [c_i]),
    "mn_plan_bytes": (c_i64, [C.POINTER(Config)]),
    "mn_create": (c_void, [C.POINTER(Config), c_void, c_void, c_void, c_void, c_void]),
    "mn_destroy": (None, [c_void]),
    "mn_set_learn_flags": (c_i, [c_void, c_i, c_i]),
    "mn_set_optim": (c_i, [c_void, c_f, c_f, c_f, c_f, c_f, c_f]),
    "mn_set_optim_method": (c_i, [c_void, c_i, c_i]),
    "mn_set_step_count": (c_i, [c_void, c_i64]),
    "mn_get_step_count": (c_i64, [c_void]),
    "mn_stuck_overflow_steps": (c_i64, [c_void]),
    "mn_set_loss_host": (c_i, [c_void, c_void]),
    "mn_wait_loss": (c_i, [c_void]),
    "mn_set_loss_scale": (c_i, [c_void, c_f, c_i]),
    "mn_get_loss_scale": (c_i, [c_void, C.POINTER(c_f), C.POINTER(c_i64)]),
    "mn_debug_tensor": (c_i, [c_void, C.c_char_p, C.POINTER(c_void), C.POINTER(c_i64), C.POINTER(C.c_int32)]),
    "mn_set_dropout": (c_i, [c_void, c_f, C.c_uint64]),
    "mn_set_dropout_calls": (c_i, [c_void, C.c_uint32]),
    "mn_set_input_u8": (c_i, [c_void, c_i, C.POINTER(c_f), C.POINTER(c_f)]),
    "mn_forward": (c_i, [c_void, c_void, c_void, c_i, c_void]),
    "mn_loss": (c_i, [c_void, c_void, c_void, c_void, c_void]),
    "mn_train_step": (c_i, [c_void, c_void, c_void, c_void, c_void, c_void]),
    "mn_train_forward_loss": (c_i, [c_void, c_void, c_void, c_void, c_void, c_void]),
    "mn_train_backward_stage": (c_i, [c_void, c_i, c_void]),
    "mn_grad_bucket": (c_i, [c_void, c_i, C.POINTER(c_i64), C.POINTER(c_i64)]),
    "mn_optim_step": (c_i, [c_void, c_f, c_void]),
    "mn_grad_bucket_pack_bf16": (c_i, [c_void, c_i, c_void, c_void]),
    "mn_grad_bucket_unpack_bf16": (c_i, [c_void, c_i, c_void, c_void]),
    "mn_params_changed": (c_i, [c_void]),
    "mn_set_profiling": (c_i, [c_void, c_i]),
    "mn_last_kernel_ms": (c_i, [c_void, c_i, C.POINTER(c_f), C.POINTER(c_i)]),
    "mn_op_igemm": (c_i, [c_i, C.POINTER(GatherGeom), c_void, c_void, c_void, c_i, c_void, c_void, c_i, c_void, c_void,
                          c_f, c_void, c_void]),
    "mn_op_igemm_grid_m": (c_i, [c_i]),
    "mn_op_wgrad": (c_i, [c_i, C.POINTER(GatherGeom), c_void, c_i, c_void, c_void, c_i, c_void, c_f, c_i, c_void, c_void]),
    "mn_op_wgrad_ws_floats": (c_i64, []),
    "mn_op_wgrad_ws": (c_i, [c_i, C.POINTER(GatherGeom), c_void, c_i, c_void, c_void, c_i, c_f, c_void, c_i64, c_void, c_void]),
    "mn_op_conv_halo_pp": (c_i, [C.POINTER(GatherGeom), c_void, c_void, c_void, c_i, c_void, c_i, c_i, c_void, c_void, c_void, c_f,
                                 c_i, c_void]),
    "mn_op_conv_halo_h2": (c_i, [C.POINTER(GatherGeom), c_void, c_void, c_void, c_i, c_void, c_i, c_void]),
    "mn_op_conv_dgrad": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_void, c_void, c_void, c_void, c_void, c_void, c_i,
                               c_void, c_void]),
    "mn_op_stem_conv": (c_i, [c_void, c_void, c_void, c_void, c_i, c_i, c_i, c_i, c_i, c_void]),
    "mn_op_stem_conv_x3": (c_i, [c_void, c_void, c_void, c_void, c_i, c_i, c_i, c_i, c_i, c_void]),
    "mn_op_dense": (c_i, [c_void, c_void, c_void, c_void, c_i, c_i, c_i, c_i, c_void]),
    "mn_op_dense_wgrad": (c_i, [c_void, c_void, c_void, c_void, c_i, c_i, c_i, c_f, c_void]),
    "mn_op_head_wgrad": (c_i, [c_void, c_void, c_void, c_void, c_void, c_void, c_i, c_i, c_f, c_i, c_void]),
    "mn_op_stem_bwd": (c_i, [c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_i, c_void, c_void, c_void,
                             c_void, c_void, c_i, c_i, c_i, c_i, c_f, c_void]),
    "mn_op_oihw_to_ohwi": (c_i, [c_void, c_void, c_i, c_i, c_i, c_i, c_i, c_void]),
    "mn_op_criterion": (c_i, [c_i, c_i, c_i, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_f, c_void]),
    "mn_op_calc_vos": (c_i, [c_void, c_i, c_i, c_void, c_void, c_void, c_void]),
    "mn_pgo_optimize": (c_i, [c_void, c_void, c_void, c_void, c_i, c_i, c_i, C.c_double, C.c_double, C.c_double,
                              C.c_double, c_i, c_void]),
    "mn_op_adam": (c_i, [c_void, c_void, c_void, c_void, c_i64, c_i64, c_f, c_f, c_f, c_f, c_f, c_i64, c_f, c_f, c_void,
                         c_i, c_void]),
    "mn_op_optim": (c_i, [c_i, c_i, c_void, c_void, c_void, c_void, c_i64, c_i64, c_f, c_f, c_f, c_f, c_f, c_i64, c_f, c_f, c_void,
                          c_i, c_void]),
    "mn_op_bn_train_fwd": (c_i, [c_i, c_void, c_i64, c_i, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_i,
                                 c_void, c_f, c_f, c_void, c_void]),
    "mn_op_bn_bwd": (c_i, [c_i, c_void, c_void, c_void, c_i64, c_i, c_void, c_void, c_void, c_void, c_void, c_void, c_void,
                           c_void, c_f, c_void]),
    "mn_op_maxpool_fwd": (c_i, [c_i, c_void, c_void, c_void, c_i, c_i, c_i, c_i, c_void]),
    "mn_op_maxpool_bwd": (c_i, [c_i, c_void, c_void, c_void, c_i, c_i, c_i, c_i, c_void]),
    "mn_op_occupy": (c_i, [c_i, c_i, c_f, c_void, c_void, c_i64, c_i, c_void]),
}

SYMBOLS = tuple(_PROTOS)


class MapNetHipError(RuntimeError):
    pass


class Binding:
    """Typed view of a shared library exporting the mapnet_hip.h ABI."""

    def __init__(self, cdll):
        self.cdll = cdll
        for name, (res, args) in _PROTOS.items():
            fn = getattr(cdll, name)  # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
            setattr(self, name[3:], fn)
        self.backend_name = self.backend().decode()

    def check(self, rc):
        if rc != 0:
            raise MapNetHipError(self.last_error().decode() or "mapnet_hip call failed (rc=%d)" % rc)

    def entries(self, feat_dim):
        out = []
        for i in range(self.model_entries(feat_dim)):
            e = Entry()
            self.check(self.model_entry(feat_dim, i, C.byref(e)))
            out.append(e)
        return out


_hip = None


def hip():
    """The product binding: libmapnet_hip.so or an exception."""
    global _hip
    if _hip is None:
        alt = os.environ.get("MN_LIB")  # measurement builds of the SAME library (tools/ablation/*.so); must be a hip backend
        if alt:
            import torch  # noqa: F401
            _hip = Binding(C.CDLL(alt))
            if _hip.backend_name != "hip":
                raise MapNetHipError("unexpected backend %r in %s" % (_hip.backend_name, alt))
            return _hip
        if not os.path.isfile(LIB_PATH):
            raise MapNetHipError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                 "(make -C geomapnet_amd/csrc); there is no CPU fallback" % LIB_PATH)
        import torch  # noqa: F401  (loads the HIP runtime the library is linked against)
        _hip = Binding(C.CDLL(LIB_PATH))
        if _hip.backend_name != "hip":
            raise MapNetHipError("unexpected backend %r in %s" % (_hip.backend_name, LIB_PATH))
    return _hip


def ptr(t):
    """device/host pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())
