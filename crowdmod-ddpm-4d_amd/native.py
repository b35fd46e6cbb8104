"""ctypes binding of libcrowdmod_hip.so (C ABI: include/crowdmod_hip.h).

The product path has no CPU fallback: if the shared library is missing or a
call fails, an exception is raised -- nothing here (or anywhere under
crowdmod-ddpm-4d_amd/) imports the oracle.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CM_LIB_PATH") or os.path.join(_HERE, "libcrowdmod_hip.so")  # override: A/B of two builds
MAX_LEVELS = 8
ABI_VERSION = 3   # CM_ABI_VERSION of include/crowdmod_hip.h


class NativeError(RuntimeError):
    """A libcrowdmod_hip call returned non-zero (message from cm_last_error)."""


class cm_unet_config(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32), ("out_channels", C.c_int32), ("num_res_blocks", C.c_int32),
        ("base_channels", C.c_int32), ("n_levels", C.c_int32),
        ("channel_mult", C.c_int32 * MAX_LEVELS), ("apply_attention", C.c_int32 * MAX_LEVELS),
        ("time_multiple", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32),
        ("past_len", C.c_int32), ("future_len", C.c_int32), ("max_batch", C.c_int32), ("device", C.c_int32),
    ]


class cm_sample_opts(C.Structure):
    _fields_ = [
        ("sampler", C.c_int32), ("guidance", C.c_int32), ("lambda_guidance", C.c_float),
        ("ddim_sigma", C.c_float), ("ddim_divider", C.c_int32), ("first_steps", C.c_int32),
        ("seed", C.c_uint64), ("sample_id_base", C.c_int64), ("use_graph", C.c_int32), ("check_finite", C.c_int32),
        ("fm_steps", C.c_int32), ("fm_time_max_pos", C.c_int32),
    ]


SAMPLER_DDPM, SAMPLER_DDIM, SAMPLER_FM_EULER = 0, 1, 2
PRECISION_F32, PRECISION_F16, PRECISION_F32R, PRECISION_F32X = 0, 1, 2, 3
GUIDANCE_NONE, GUIDANCE_SPARSITY = 0, 1
TABLES = ("beta", "alpha", "alpha_bar", "sqrt_alpha_bar", "one_by_sqrt_alpha", "sqrt_one_minus_alpha_bar")

_lib: Optional[C.CDLL] = None

# name -> (restype, argtypes); the exported symbol list of include/crowdmod_hip.h
_P = C.c_void_p
_F = C.POINTER(C.c_float)
SIGNATURES = {
    "cm_last_error": (C.c_char_p, []),
    "cm_abi_version": (C.c_int, []),
    "cm_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "cm_malloc": (C.c_int, [C.c_int, C.POINTER(_P), C.c_size_t]),
    "cm_free": (C.c_int, [C.c_int, _P]),
    "cm_memcpy_h2d": (C.c_int, [C.c_int, _P, _P, C.c_size_t]),
    "cm_memcpy_d2h": (C.c_int, [C.c_int, _P, _P, C.c_size_t]),
    "cm_memcpy_d2d": (C.c_int, [C.c_int, _P, _P, C.c_size_t]),
    "cm_device_synchronize": (C.c_int, [C.c_int]),
    "cm_model_create": (C.c_int, [C.POINTER(cm_unet_config), C.POINTER(_P)]),
    "cm_model_destroy": (C.c_int, [_P]),
    "cm_model_num_params": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "cm_model_param_info": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "cm_model_set_param": (C.c_int, [_P, C.c_char_p, _P, C.c_int64]),
    "cm_model_get_param": (C.c_int, [_P, C.c_char_p, _P, C.c_int64]),
    "cm_model_set_precision": (C.c_int, [_P, C.c_int32]),
    "cm_model_finalize": (C.c_int, [_P]),
    "cm_unet_forward": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, _P]),
    "cm_unet_forward_host": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32]),
    "cm_model_dropout_width": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "cm_unet_forward_train": (C.c_int, [_P, _P, _P, _P, _P, C.c_float, C.c_uint64, C.c_int64, _P, C.c_int32, _P]),
    "cm_mse_loss": (C.c_int, [_P, _P, _P, C.c_int64, C.POINTER(C.c_float), _P]),
    "cm_debug_activation": (C.c_int, [_P, C.c_char_p, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "cm_schedule_create": (C.c_int, [C.c_int32, C.c_float, C.c_float, C.c_float, C.c_int32, C.POINTER(_P)]),
    "cm_schedule_destroy": (C.c_int, [_P]),
    "cm_schedule_table": (C.c_int, [_P, C.c_int32, _P, C.c_int32]),
    "cm_q_sample": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int64, _P]),
    "cm_ddpm_step": (C.c_int, [_P, _P, _P, C.c_int32, _P, C.c_uint64, C.c_int64, C.c_int32, C.c_int64, _P]),
    "cm_sample_loop": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(cm_sample_opts), _P, _P, C.c_int32, _P]),
    "cm_sample_loop_host": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(cm_sample_opts), _P, _P, C.c_int32]),
    "cm_sample_num_steps": (C.c_int, [_P, C.POINTER(cm_sample_opts), C.POINTER(C.c_int32)]),
    "cm_profile_enable": (C.c_int, [_P, C.c_int32]),
    "cm_profile_read": (C.c_int, [_P, _F, C.POINTER(C.c_int64)]),
    "cm_profile_read_union": (C.c_int, [_P, _F]),
    "cm_profile_report": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "cm_model_cost": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "cm_model_class_flops": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_double)]),
    "cm_model_exec_flops": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_double)]),
    "cm_model_issue_flops": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "cm_frame_metrics": (C.c_int, [C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "cm_debug_conv_flags": (C.c_int, [C.c_int32]),
    "cm_debug_conv_count": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "cm_debug_conv_info": (C.c_int, [_P, C.c_int32, C.c_char_p, C.c_int64]),
    "cm_debug_conv_io": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, C.c_int32]),
    "cm_debug_time_conv": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.POINTER(C.c_float)]),
    "cm_train_init": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]),
    "cm_train_set_lr": (C.c_int, [_P, C.c_float]),
    "cm_train_set_sample_base": (C.c_int, [_P, C.c_int64]),
    "cm_train_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_uint64, C.POINTER(C.c_float), C.c_int32, C.c_int32, _P]),
    "cm_train_step_xt": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_uint64, C.POINTER(C.c_float), C.c_int32, C.c_int32, _P]),
    "cm_train_get_grad": (C.c_int, [_P, C.c_char_p, _P, C.c_int64]),
    "cm_train_sync": (C.c_int, [_P]),
    "cm_train_flat_grads": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "cm_train_apply": (C.c_int, [_P, _P]),
    "cm_train_get_opt_state": (C.c_int, [_P, C.c_char_p, C.c_int32, _P, C.c_int64]),
    "cm_train_set_opt_state": (C.c_int, [_P, C.c_char_p, C.c_int32, _P, C.c_int64]),
    "cm_train_opt_step": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int32]),
}


def _preload_hip_runtime():
    """Make this process use ONE HIP runtime.

    PyTorch-ROCm bundles its own libamdhip64.so (same SONAME as /opt/rocm's).  If
    our library pulled in the system copy and torch later loaded its own, two HIP
    runtimes would share the process.  Loading torch's copy first (by path, without
    importing torch) lets the dynamic loader resolve our DT_NEEDED to it too.
    """
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raise if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C crowdmod-ddpm-4d_amd/csrc`). There is no CPU fallback.")
    _preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    if L.cm_abi_version() != ABI_VERSION:
        raise NativeError("libcrowdmod_hip.so ABI version mismatch")
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().cm_last_error()
        raise NativeError(msg.decode("utf-8", "replace") if msg else f"libcrowdmod_hip call failed ({rc})")


def device_count() -> int:
    n = C.c_int(0)
    check(lib().cm_device_count(C.byref(n)))
    return n.value


def ptr(x) -> int:
    """Device or host address of a numpy array / torch tensor / DeviceBuffer / int / None."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    raise TypeError(f"cannot take the address of {type(x)}")


class DeviceBuffer:
    """Raw device allocation owned by Python (cm_malloc / cm_free)."""

    def __init__(self, nbytes: int, device: int = 0):
        self.device, self.nbytes = device, int(nbytes)
        p = C.c_void_p()
        check(lib().cm_malloc(device, C.byref(p), self.nbytes))
        self.ptr = p.value

    def upload(self, arr: np.ndarray) -> "DeviceBuffer":
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(lib().cm_memcpy_h2d(self.device, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, shape, dtype=np.float32) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(lib().cm_memcpy_d2h(self.device, out.ctypes.data, self.ptr, out.nbytes))
        return out

    @classmethod
    def from_array(cls, arr: np.ndarray, device: int = 0) -> "DeviceBuffer":
        arr = np.ascontiguousarray(arr)
        return cls(arr.nbytes, device).upload(arr)

    def free(self):
        if getattr(self, "ptr", None):
            lib().cm_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
