"""ctypes binding of the C-ABI library (include/picklebot_amd.h).

The product path has NO fallback: if the HIP library is missing or a call fails, a RuntimeError
is raised.  PyTorch only supplies device memory, the current HIP stream and the device ordinal.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpicklebot_amd.so")

PB_F32, PB_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_LEAKY_RELU, ACT_HSWISH, ACT_HSIGMOID, ACT_SILU = range(6)
ABI_VERSION = 21
STAT_SLOTS = 512


class DwShape(Structure):
    _fields_ = [(n, c_int) for n in ("N", "C", "Ti", "Hi", "Wi", "To", "Ho", "Wo", "kt", "kh", "kw",
                                     "st", "sh", "sw", "pt", "ph", "pw")]


class ConvShape(Structure):
    _fields_ = [(n, c_int) for n in ("N", "Cin", "Cout", "Ti", "Hi", "Wi", "To", "Ho", "Wo", "kt", "kh", "kw",
                                     "st", "sh", "sw", "pt", "ph", "pw")]


class ReduceJob(Structure):
    _fields_ = [("part", c_void_p), ("out", c_void_p), ("n", c_int64), ("stride", c_int64), ("split", c_int),
                ("accumulate", c_int)]


class StemShape(Structure):
    _fields_ = ([(n, c_int) for n in ("N", "Cin", "Cout", "Ti", "Hi", "Wi", "To", "Ho", "Wo", "kt", "kh", "kw",
                                      "st", "sh", "sw", "pt", "ph", "pw")] +
                [(n, c_int64) for n in ("xs_n", "xs_c", "xs_t", "xs_h", "xs_w")])


P, F, I, L, D = c_void_p, c_float, c_int, c_int64, c_double

# name -> argument types (every entry returns int status except the two noted)
SIGNATURES = {
    "pb_pwconv_fwd": [P, P, P, L, I, I, I, I, I, P, P, I, F, P, L, P, P, I, F, P, I, P],
    "pb_partials_reduce": [POINTER(ReduceJob), I, I, P],
    "pb_pwconv_bwd_weight_ws": [P, P, P, L, I, I, I, P, P, I, F, P, L, P, L, POINTER(ReduceJob), I, P],
    "pb_dwconv3d_fwd": [P, P, P, POINTER(DwShape), I, P, I, I, P],
    "pb_dwconv3d_bwd_data": [P, P, P, POINTER(DwShape), I, P, P, I, P],
    "pb_dwconv3d_fwd_hist": [P, P, I, P, P, POINTER(DwShape), I, I, P],
    "pb_dwconv3d_bwd_weight": [P, P, P, POINTER(DwShape), I, P, P, P, L, POINTER(ReduceJob), I, P],
    "pb_dwconv3d_bwd_fused": [P, P, P, P, P, POINTER(DwShape), I, P, P, P, L, POINTER(ReduceJob), I, P],
    "pb_frames_gather": [P, P, I, I, I, L, I, I, I, P, I, I, P],
    "pb_stem_conv3d_fwd": [P, I, F, P, P, P, POINTER(StemShape), I, P, I, I, P],
    "pb_stem_conv3d_bwd_weight": [P, P, I, F, P, P, POINTER(StemShape), I, P, L, I, P],
    "pb_chan_stats": [P, L, I, I, P, I, P],
    "pb_bn_finalize": [P, I, D, P, P, F, F, P, P, P, I, P, P, P, P, I, I, P],
    "pb_bn_fold_batch": [P, I, I, P],
    "pb_affine_act_fwd": [P, P, L, I, I, P, P, P, I, F, P, P, L, P, L, L, L, I, P],
    "pb_act_bwd_reduce": [P, P, P, L, I, I, P, P, I, F, P, L, L, F, P, P, P, L, L, L, I, P],
    "pb_bn_bwd_finalize": [P, I, D, P, P, P, I, P, P, P, P, P, I, I, P],
    "pb_affine2": [P, P, P, L, I, I, P, P, P, P, P, L, P, P, I, F, I, P],
    "pb_rowgroup_reduce": [P, P, P, L, L, I, I, P, L, I, P],
    "pb_se_fc_fwd": [P, F, P, P, P, P, P, P, I, I, I, I, P],
    "pb_se_fc_bwd": [P, P, F, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P],
    "pb_se_fused_fwd": [P, I, F, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P],
    "pb_se_fused_bwd": [P, I, P, F, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P],
    "pb_se_fc_bwd_w": [P, P, F, P, P, P, P, P, I, I, I, I, P],
    "pb_pool_head_fwd": [P, I, F, P, P, P, P, I, P, P, P, P, I, I, I, I, I, P],
    "pb_pool_head_bwd": [P, F, P, P, P, I, P, P, P, I, I, I, I, I, P],
    "pb_pool_head_bwd_w": [P, P, F, P, I, P, P, P, P, I, I, I, I, I, P],
    "pb_rowgroup_partials": [P, P, P, L, L, I, I, I, P],
    "pb_transpose_f32": [P, P, I, I, I, P],
    "pb_pw_weight_prep": [P, I, I, I, P, I, P, I, P],
    "pb_pw_weight_prep_batch": [P, I, I, P],
    "pb_pwconv_fwd_bf16w": [P, P, P, L, I, I, I, L, P, P, I, F, P, P, I, P],
    "pb_pwconv_fwd_fold4": [P, P, P, L, I, I, F, P, I, P],
    "pb_unfold_fold": [P, P, I, I, I, I, I, I, I, I, I, I, I, P],
    "pb_layernorm": [P, P, P, P, L, I, F, I, I, I, P],
    "pb_copy_cols": [P, L, I, P, L, I, L, I, I, I, I, P],
    "pb_layernorm_wide": [P, P, P, P, P, P, L, I, F, I, I, P],
    "pb_stream_add": [P, I, P, P, P, L, I, P],
    "pb_attention_fwd": [P, P, P, L, I, F, I, I, P],
    "pb_attention_fwd_fp8": [P, P, P, L, I, F, I, P],
    "pb_attention_bwd": [P, P, P, P, P, P, L, I, F, I, I, P],
    "pb_linattn_fwd": [P, P, P, P, I, I, I, I, I, P],
    "pb_linattn_bwd": [P, P, P, P, P, P, P, I, I, I, I, I, P],
    "pb_conv_weight_prep": [P, P, I, I, I, I, I, P],
    "pb_conv3d_fwd": [P, P, P, POINTER(ConvShape), I, P, P, I, F, P, I, P],
    "pb_conv3d_fwd_bf16w": [P, P, P, POINTER(ConvShape), P, P, P, I, F, P, I, P],
    "pb_adamw_flat": [P, P, P, P, L, F, F, F, F, F, P, P, P, P, I, P],
    "pb_grad_check": [P, L, P, I, P],
    "pb_conv3d_bwd_weight_ws": [P, P, P, POINTER(ConvShape), I, P, L, P, I, P],
}

_lib = None


# entry points returning a static C string instead of a status
STRING_SIGNATURES = {
    "pb_dwconv3d_kernel_name": [POINTER(DwShape), I, I],
}

# entry points returning a size (int64) instead of a status
SIZE_SIGNATURES = {
    "pb_pwconv_bwd_weight_ws_bytes": [L, I, I, I, I, I, L, I],
    "pb_conv3d_bwd_weight_ws_bytes": [POINTER(ConvShape), I, I],
    "pb_dwconv3d_bwd_weight_ws_bytes": [POINTER(DwShape), I],
    "pb_dwconv3d_bwd_fused_ws_bytes": [POINTER(DwShape), I],
    "pb_stem_conv3d_bwd_weight_ws_bytes": [POINTER(StemShape), I, F, I],
    "pb_rowgroup_reduce_ws_bytes": [L, L, I, I],
}

# entry points returning a plain int count
COUNT_SIGNATURES = {
    "pb_dwconv3d_pool_slots": [POINTER(DwShape), I],
    "pb_rowgroup_slots": [L, L, I, I],
    "pb_pwconv_fwd_bf16w_path": [L, I, I, I, L, I],
}


def load():
    """Load (once) and type the library.  Raises RuntimeError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"picklebot_amd: HIP library not built ({LIB_PATH}); run `python -m picklebot_amd.build` "
            "(there is no CPU/eager fallback)")
    lib = ctypes.CDLL(LIB_PATH)
    lib.pb_last_error.restype = c_char_p
    lib.pb_last_error.argtypes = []
    lib.pb_abi_version.restype = c_int
    lib.pb_abi_version.argtypes = []
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = c_int
        fn.argtypes = args
    for name, args in STRING_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = c_char_p
        fn.argtypes = args
    for name, args in SIZE_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int64
        fn.argtypes = args
    for name, args in COUNT_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = c_int
        fn.argtypes = args
    if lib.pb_abi_version() != ABI_VERSION:
        raise RuntimeError(f"picklebot_amd: ABI version mismatch ({lib.pb_abi_version()} != {ABI_VERSION}); rebuild")
    _lib = lib
    return lib


def call(name, *args, allow=()):
    """Run an entry point; a status listed in `allow` is returned to the caller instead of raising."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc in allow:
        return rc
    if rc != 0:
        raise RuntimeError(f"{name} failed (status {rc}): {lib.pb_last_error().decode()}")
