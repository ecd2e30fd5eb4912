"""ctypes binding of libvae_amd.so (C ABI: include/vae_amd.h).

There is no CPU fallback: if the library is missing or the device is not an MI355X the
import of the first op raises.  torch is used only for device memory and streams.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VAE_AMD_LIB") or os.path.join(_HERE, "csrc", "libvae_amd.so")  # (override: instrumented builds)

# enums (include/vae_amd.h)
OP_CONV, OP_CONVT, OP_BN, OP_LINEAR, OP_REPARAM, OP_SLICE0, OP_ELBO = 1, 2, 3, 4, 5, 6, 7
ACT_NONE, ACT_LEAKY, ACT_SIGMOID = 0, 1, 2
F_ACC_DIN, F_BN_RELU, F_KL_GRAD, F_EXT_GRAD, F_BCE, F_IN_RELU, F_FUSED = 1, 2, 4, 8, 16, 32, 64
F_LOSS_IN_BWD = 128
F_BN_ACC = 256
BN_ACC_MAXC = 16  # igemm.h BNFIN_MAXC


def bn_acc_floats(C):
    return 1536 * C  # include/vae_amd.h VAE_BN_ACC_FLOATS
DP_ID_BYTES = 256  # include/vae_amd.h: VAE_DP_ID_BYTES
MAX_CHUNKS, ELBO_BLOCKS, MAX_WORLD = 64, 1024, 64
SP_WORK, SP_PARAM, SP_GRAD, SP_STATE, SP_EXT0 = 0, 1, 2, 3, 4


class VaeOp(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("act", ctypes.c_int32), ("flags", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("r", ctypes.c_int64 * 12), ("d", ctypes.c_int32 * 16),
                ("f", ctypes.c_float * 4)]


class VaeAmdError(RuntimeError):
    pass


_lib = None


def lib():
    """Loads libvae_amd.so once.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VaeAmdError(
            f"{LIB_PATH} not found: build it with `make -C vae_amd/csrc` (or __graft_entry__.build()). "
            "vae_amd has no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
    L.vae_last_error.restype = ctypes.c_char_p
    L.vae_last_error.argtypes = []
    L.vae_check_device.argtypes = []
    L.vae_run_forward.argtypes = [ctypes.POINTER(VaeOp), i32, ctypes.POINTER(vp), ctypes.POINTER(i64), i32, i32, i32, vp]
    L.vae_time_forward.argtypes = [ctypes.POINTER(VaeOp), i32, ctypes.POINTER(vp), ctypes.POINTER(i64), i32, i32, i32, vp,
                                   ctypes.POINTER(ctypes.c_float)]
    L.vae_time_seq.argtypes = [ctypes.POINTER(VaeOp), i32, ctypes.POINTER(vp), ctypes.POINTER(i64), i32, i32, vp,
                               ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double), ctypes.c_char_p, i32,
                               ctypes.POINTER(i32)]
    L.vae_time_step.argtypes = [ctypes.POINTER(VaeOp), i32, ctypes.POINTER(vp), ctypes.POINTER(i64), i32, i32, vp, vp, vp, vp, vp,
                                ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                ctypes.c_char_p, i32, ctypes.POINTER(i32)]
    L.vae_run_backward.argtypes = [ctypes.POINTER(VaeOp), i32, ctypes.POINTER(vp), ctypes.POINTER(i64), i32, i32, vp]
    L.vae_run_backward_range.argtypes = [ctypes.POINTER(VaeOp), i32, ctypes.POINTER(vp), ctypes.POINTER(i64), i32, i32, vp,
                                         i32, i32, i32]
    L.vae_set_collective.argtypes = [ALLGATHER_FN, vp, i32, i32]
    L.vae_run_backward_step.argtypes = [ctypes.POINTER(VaeOp), i32, ctypes.POINTER(vp), ctypes.POINTER(i64), i32, i32, vp,
                                        vp, vp, vp, vp, vp, i64, vp]
    L.vae_conv_fused_ok.argtypes = [ctypes.POINTER(VaeOp), i32]
    L.vae_dp_unique_id.argtypes = [ctypes.c_char_p, vp]
    L.vae_dp_init.argtypes = [ctypes.c_char_p, vp, i32, i32]
    L.vae_dp_world.argtypes = []
    L.vae_dp_enable.argtypes = [i32]
    L.vae_dp_all_gather_rows.argtypes = [vp, i64, vp]
    L.vae_dp_all_reduce_sum.argtypes = [vp, i64, vp]
    L.vae_dp_shutdown.argtypes = []
    L.vae_dp_set_single_comm.argtypes = [i32]
    L.vae_dp_single_comm.argtypes = []
    L.vae_conv_forward_variant.argtypes = [ctypes.POINTER(VaeOp), i32]
    L.vae_conv_scratch_floats.argtypes = [ctypes.POINTER(VaeOp), i32]
    L.vae_conv_scratch_floats.restype = i64
    L.vae_conv_wgrad_partial_floats.argtypes = [ctypes.POINTER(VaeOp), i32]
    L.vae_conv_wgrad_partial_floats.restype = i64
    L.vae_linear_wgrad_partial_floats.argtypes = [ctypes.POINTER(VaeOp), i32]
    L.vae_linear_wgrad_partial_floats.restype = i64
    L.vae_set_compute_dtype.argtypes = [i32]
    L.vae_get_compute_dtype.argtypes = []
    L.vae_seq_scratch_floats.argtypes = [ctypes.POINTER(VaeOp), i32, i32, i64]
    L.vae_seq_scratch_floats.restype = i64
    L.vae_elbo_forward.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i64, i64, i32, i32, vp]
    L.vae_elbo_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i64, i64, i32, i32, vp]
    L.vae_elbo_forward_noalloc.argtypes = [vp, vp, vp, vp, vp, i32, i32, i64, i64, i32, i32, vp]
    L.vae_adam_step.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, vp, vp]
    L.vae_adam_step_hyper.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp]
    L.vae_randn.argtypes = [vp, i64, vp, vp]
    L.vae_window_gather.argtypes = [vp, i32, i64, i32, vp, i32, vp, vp]
    L.vae_graph_begin.argtypes = [ctypes.POINTER(vp)]
    L.vae_graph_end.argtypes = [ctypes.POINTER(vp)]
    L.vae_graph_launch.argtypes = [vp, vp]
    L.vae_graph_destroy.argtypes = [vp]
    for name in ("vae_check_device", "vae_conv_fused_ok", "vae_conv_forward_variant", "vae_run_forward", "vae_time_forward", "vae_time_seq", "vae_set_compute_dtype", "vae_get_compute_dtype", "vae_run_backward", "vae_run_backward_range", "vae_set_collective", "vae_elbo_forward",
                 "vae_elbo_forward_noalloc", "vae_elbo_backward", "vae_adam_step", "vae_adam_step_hyper", "vae_randn", "vae_window_gather", "vae_graph_begin", "vae_graph_end",
                 "vae_graph_launch", "vae_graph_destroy", "vae_dp_unique_id", "vae_dp_init", "vae_dp_world", "vae_dp_enable",
                 "vae_dp_all_gather_rows", "vae_dp_all_reduce_sum", "vae_dp_shutdown", "vae_run_backward_step",
                 "vae_dp_set_single_comm", "vae_dp_single_comm", "vae_time_step"):
        getattr(L, name).restype = i32
    _lib = L
    return L


# int fn(void* user, float* buf, int64 floats_per_rank, void* stream)  (include/vae_amd.h: vae_allgather_fn)
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)

EXPORTS = ("vae_seq_scratch_floats", "vae_last_error", "vae_check_device", "vae_conv_fused_ok", "vae_conv_forward_variant", "vae_conv_scratch_floats", "vae_conv_wgrad_partial_floats", "vae_linear_wgrad_partial_floats", "vae_set_collective", "vae_run_forward", "vae_time_forward", "vae_time_seq", "vae_set_compute_dtype", "vae_get_compute_dtype", "vae_run_backward", "vae_run_backward_range", "vae_elbo_forward",
           "vae_elbo_forward_noalloc", "vae_elbo_backward", "vae_adam_step", "vae_adam_step_hyper", "vae_randn", "vae_window_gather", "vae_graph_begin", "vae_graph_end",
           "vae_graph_launch", "vae_graph_destroy", "vae_dp_unique_id", "vae_dp_init", "vae_dp_world", "vae_dp_enable",
           "vae_dp_all_gather_rows", "vae_dp_all_reduce_sum", "vae_dp_shutdown", "vae_run_backward_step",
           "vae_dp_set_single_comm", "vae_dp_single_comm", "vae_time_step")


_DTYPES = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}


def set_compute_dtype(dtype):
    """"f32" (default) or "bf16": the dtype of the matrix kernels' operands (include/vae_amd.h,
    vae_set_compute_dtype: bf16 operands, fp32 accumulate, fp32 storage).  Process-global; programs and fused steppers
    are cached per dtype, a stepper refuses to run under a dtype other than the one it was built for."""
    key = str(dtype).replace("torch.", "")
    if key not in _DTYPES:
        raise ValueError(f"compute dtype {dtype!r}: expected 'f32' or 'bf16'")
    check(lib().vae_set_compute_dtype(_DTYPES[key]), "vae_set_compute_dtype")


def get_compute_dtype():
    return "bf16" if lib().vae_get_compute_dtype() == 1 else "f32"


def check(rc, what):
    if rc != 0:
        raise VaeAmdError(f"{what} failed ({rc}): {lib().vae_last_error().decode()}")


_device_ok = set()


def require_device(t):
    """Every op calls this on one of its tensors: product code runs on an MI355X or not at all."""
    if not t.is_cuda:
        raise VaeAmdError(f"vae_amd ops need tensors on an MI355X (got device {t.device}); there is no CPU path")
    idx = t.device.index if t.device.index is not None else torch.cuda.current_device()
    if idx not in _device_ok:
        with torch.cuda.device(idx):
            check(lib().vae_check_device(), "vae_check_device")
        _device_ok.add(idx)


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
