"""ctypes binding of libccv_mppi_hip.so -- exactly the symbols include/ccv_mppi.h and
include/ccv_mppi_host.h declare.  No torch types cross this boundary; there is no CPU fallback:
if the library is missing it is built with hipcc, and if that fails the import raises.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

MAX_UDIM = 5
MAX_HORIZON = 128
ABI_VERSION = 1

OK = 0
ERR_INVALID_ARG = -1
ERR_NO_DEVICE = -2
ERR_HIP = -3
ERR_STATE = -4
ERR_ALLOC = -5
ERR_TIMEOUT = -6

DIFF_DRIVE, STEERING_DIFF_DRIVE, FULL_BODY = 0, 1, 2
FLAG_ROLL_OFF, FLAG_STEER_OFF, FLAG_MIN_SHIFT, FLAG_NO_STATE_STORE = 0x1, 0x2, 0x4, 0x8


class Config(C.Structure):
    """struct ccv_mppi_config"""
    _fields_ = [
        ("abi_version", C.c_int32), ("model", C.c_int32), ("num_samples", C.c_int32), ("horizon", C.c_int32),
        ("sample_offset", C.c_int32), ("device", C.c_int32), ("flags", C.c_int32), ("reserved", C.c_int32),
        ("control_noise", C.c_double), ("lam", C.c_double), ("v_ref", C.c_double),
        ("u_min", C.c_double * MAX_UDIM), ("u_max", C.c_double * MAX_UDIM),
        ("path_weight", C.c_double), ("v_weight", C.c_double), ("zmp_weight", C.c_double),
        ("roll_v_weight", C.c_double), ("back_weight", C.c_double), ("yaw_weight", C.c_double),
    ]


class Stats(C.Structure):
    """struct ccv_mppi_stats"""
    _fields_ = [
        ("sum_w", C.c_double), ("min_cost", C.c_double), ("max_cost", C.c_double),
        ("n_zero_weight", C.c_int64), ("nonfinite", C.c_int32), ("reserved", C.c_int32),
        ("device_us", C.c_float), ("rollout_us", C.c_float),
    ]


_dp = C.POINTER(C.c_double)
_H = C.c_void_p

# name -> (restype, argtypes): every symbol of the two public headers
SIGNATURES = {
    "ccv_mppi_create": (C.c_int, [C.POINTER(Config), C.POINTER(_H)]),
    "ccv_mppi_destroy": (C.c_int, [_H]),
    "ccv_mppi_set_stream": (C.c_int, [_H, C.c_void_p]),
    "ccv_mppi_last_error": (C.c_char_p, [_H]),
    "ccv_mppi_version": (C.c_char_p, []),
    "ccv_mppi_udim": (C.c_int, [C.c_int]),
    "ccv_mppi_set_nominal": (C.c_int, [_H, _dp]),
    "ccv_mppi_get_nominal": (C.c_int, [_H, _dp]),
    "ccv_mppi_iterate": (C.c_int, [_H, _dp, C.c_double, _dp, _dp, C.c_double, C.c_uint64, C.c_uint64, _dp,
                                   C.POINTER(Stats)]),
    "ccv_mppi_iterate_enqueue": (C.c_int, [_H, _dp, C.c_double, _dp, _dp, C.c_double, C.c_uint64, C.c_uint64]),
    "ccv_mppi_synchronize": (C.c_int, [_H]),
    "ccv_mppi_sample": (C.c_int, [_H, C.c_uint64, C.c_uint64]),
    "ccv_mppi_inject_controls": (C.c_int, [_H, _dp]),
    "ccv_mppi_rollout": (C.c_int, [_H, _dp, C.c_double]),
    "ccv_mppi_weights": (C.c_int, [_H, _dp, _dp, C.c_double]),
    "ccv_mppi_update": (C.c_int, [_H, _dp, C.POINTER(Stats)]),
    "ccv_mppi_read_candidates": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, _dp]),
    "ccv_mppi_read_top_candidates": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_int32), _dp, _dp]),
    "ccv_mppi_read_costs": (C.c_int, [_H, C.c_int32, C.c_int32, _dp]),
    "ccv_mppi_read_weights": (C.c_int, [_H, C.c_int32, C.c_int32, _dp]),
    "ccv_mppi_read_controls": (C.c_int, [_H, C.c_int32, C.c_int32, _dp]),
    "ccv_mppi_partials_size": (C.c_int, [_H]),
    "ccv_mppi_iterate_partials_enqueue": (C.c_int, [_H, _dp, C.c_double, _dp, _dp, C.c_double, C.c_uint64,
                                                    C.c_uint64, C.c_void_p]),
    "ccv_mppi_apply_partials_enqueue": (C.c_int, [_H, C.c_void_p]),
    "ccv_mppi_exchange_handle_bytes": (C.c_int, []),
    "ccv_mppi_exchange_create": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_void_p]),
    "ccv_mppi_exchange_connect": (C.c_int, [_H, C.c_void_p]),
    "ccv_mppi_exchange_info": (C.c_int, [_H, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ccv_mppi_iterate_exchange_enqueue": (C.c_int, [_H, _dp, C.c_double, _dp, _dp, C.c_double, C.c_uint64, C.c_uint64]),
    "ccv_mppi_resident_set_path": (C.c_int, [_H, _dp, _dp, C.c_int32, C.c_double]),
    "ccv_mppi_resident_set_pose": (C.c_int, [_H, _dp]),
    "ccv_mppi_resident_step_enqueue": (C.c_int, [_H, C.c_double, C.c_uint64, C.c_uint64, C.c_int32]),
    "ccv_mppi_resident_step_partials_enqueue": (C.c_int, [_H, C.c_double, C.c_uint64, C.c_uint64, C.c_int32, C.c_void_p]),
    "ccv_mppi_resident_step_exchange_enqueue": (C.c_int, [_H, C.c_double, C.c_uint64, C.c_uint64, C.c_int32]),
    "ccv_mppi_resident_read": (C.c_int, [_H, _dp, C.POINTER(C.c_int32), _dp, _dp, _dp, C.POINTER(C.c_int64)]),
    "ccv_mppi_resident_read_trace": (C.c_int, [_H, C.c_int32, _dp, C.POINTER(C.c_int32)]),
    "ccv_mppi_timing_enable": (C.c_int, [_H, C.c_int32]),
    "ccv_mppi_timing_read": (C.c_int, [_H, _dp, _dp, C.POINTER(C.c_int64), C.c_int32]),
    # include/ccv_mppi_host.h
    "ccv_mppi_calc_ref_path": (C.c_int, [_dp, _dp, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double,
                                         C.c_double, C.c_int32, _dp, _dp, _dp]),
    "ccv_mppi_path_cosine": (C.c_int, [_dp, _dp, _dp, C.c_double, C.c_double, C.c_double, C.c_double, _dp, _dp,
                                       C.c_int32]),
    "ccv_mppi_path_dkan": (C.c_int, [C.c_double, _dp, _dp, C.c_int32]),
    "ccv_mppi_plant_step": (C.c_int, [C.c_int32, _dp, _dp, C.c_double]),
    # include/ccv_mppi_node.hpp (extern "C" part): the ROS-free mirror of the reference controller classes
    "ccv_mppi_node_create": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), _dp, C.c_int, C.c_int, C.POINTER(_H)]),
    "ccv_mppi_node_destroy": (C.c_int, [_H]),
    "ccv_mppi_node_set_path": (C.c_int, [_H, _dp, _dp, C.c_int]),
    "ccv_mppi_node_set_state": (C.c_int, [_H, _dp]),
    "ccv_mppi_node_set_seed": (C.c_int, [_H, C.c_uint64]),
    "ccv_mppi_node_set_fused": (C.c_int, [_H, C.c_int]),
    "ccv_mppi_node_set_device_prologue": (C.c_int, [_H, C.c_int]),
    "ccv_mppi_node_run_once": (C.c_int, [_H, C.c_double, _dp]),
    "ccv_mppi_node_get_optimal": (C.c_int, [_H, _dp]),
    "ccv_mppi_node_get_ref_path": (C.c_int, [_H, _dp]),
    "ccv_mppi_node_get_optimal_path": (C.c_int, [_H, _dp]),
    "ccv_mppi_node_fb_imu": (C.c_int, [_H, _dp, _dp, _dp, _dp]),
    "ccv_mppi_node_fb_wrench": (C.c_int, [_H, C.c_int, _dp, _dp]),
    "ccv_mppi_node_fb_pose": (C.c_int, [_H, C.c_double, C.c_double, C.c_double]),
    "ccv_mppi_node_fb_update_state": (C.c_int, [_H, C.c_double]),
    "ccv_mppi_node_fb_read": (C.c_int, [_H, _dp]),
    "ccv_mppi_fb_estimator_create": (C.c_int, [C.POINTER(_H)]),
    "ccv_mppi_fb_estimator_destroy": (C.c_int, [_H]),
    "ccv_mppi_fb_estimator_imu": (C.c_int, [_H, _dp, _dp, _dp, _dp]),
    "ccv_mppi_fb_estimator_wrench": (C.c_int, [_H, C.c_int, _dp, _dp]),
    "ccv_mppi_fb_estimator_update": (C.c_int, [_H, C.c_double, C.c_double, C.c_double, C.c_double]),
    "ccv_mppi_fb_estimator_read": (C.c_int, [_H, _dp]),
}

_lib = None


def load():
    """dlopen the in-tree library (building it first if it is missing or stale)."""
    global _lib
    if _lib is None:
        path = os.environ.get("CCV_MPPI_LIB") or _build.build()   # CCV_MPPI_LIB: experiment builds only
        if not os.path.exists(path):
            raise ImportError("libccv_mppi_hip.so is missing and could not be built; there is no CPU fallback")
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError = the .so does not export what the header declares
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def dptr(a):
    return a.ctypes.data_as(_dp)


def as_f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("expected shape %s, got %s" % (shape, a.shape))
    return a
