"""ctypes binding of oracle/_build/libmppi_oracle.so (TEST INFRASTRUCTURE ONLY).

The shared library is the C++ restatement of the reference's hot path
(mppi_oracle.cpp; every function there cites the reference file:line it follows).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmppi_oracle.so")

MODELS = {"diff_drive": 0, "steering_diff_drive": 1, "full_body": 2}
UDIM = {0: 2, 1: 3, 2: 5}


class OrcConfig(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("K", C.c_int32), ("H", C.c_int32),
        ("roll_off", C.c_int32), ("steer_off", C.c_int32), ("pad_", C.c_int32),
        ("sigma", C.c_double), ("lam", C.c_double), ("v_ref", C.c_double),
        ("u_min", C.c_double * 5), ("u_max", C.c_double * 5),
        ("path_weight", C.c_double), ("v_weight", C.c_double), ("zmp_weight", C.c_double),
        ("roll_v_weight", C.c_double), ("back_weight", C.c_double), ("yaw_weight", C.c_double),
    ]


def build(force=False):
    """Compile the oracle with g++ (make).  Building the checker is not using it."""
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
            for f in ("mppi_oracle.cpp", "philox_normal.h")):
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(OrcConfig)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_udim.argtypes = [C.c_void_p]
        for name in ("orc_set_nominal", "orc_get_nominal", "orc_set_controls", "orc_get_controls",
                     "orc_get_costs", "orc_get_weights"):
            getattr(L, name).argtypes = [C.c_void_p, dp]
        L.orc_sampling_mt19937.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_sampling_philox.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32]
        L.orc_predict_states.argtypes = [C.c_void_p, dp, C.c_double]
        L.orc_calc_weights.argtypes = [C.c_void_p, dp, dp, C.c_double]
        L.orc_determine_optimal.argtypes = [C.c_void_p]
        L.orc_set_by_value.argtypes = [C.c_void_p, C.c_int]
        L.orc_fbest_create.restype = C.c_void_p
        L.orc_fbest_destroy.argtypes = [C.c_void_p]
        L.orc_fbest_imu.argtypes = [C.c_void_p, dp, dp, dp, dp]
        L.orc_fbest_wrench.argtypes = [C.c_void_p, C.c_int, dp, dp]
        L.orc_fbest_true_zmp.argtypes = [C.c_void_p]
        L.orc_fbest_true_zmp.restype = C.c_int
        L.orc_fbest_state.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double]
        L.orc_fbest_read.argtypes = [C.c_void_p, dp]
        L.orc_get_sum_w.restype = C.c_double
        L.orc_get_sum_w.argtypes = [C.c_void_p]
        L.orc_get_states.argtypes = [C.c_void_p, C.c_int, dp]
        L.orc_iterate.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, dp, C.c_double, dp, dp,
                                  C.c_double, dp]
        L.orc_calc_ref_path.restype = C.c_int
        L.orc_calc_ref_path.argtypes = [dp, dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                        C.c_int, dp, dp, dp]
        L.orc_path_cosine.restype = C.c_int
        L.orc_path_cosine.argtypes = [C.c_double] * 13 + [dp, dp, C.c_int]
        L.orc_path_dkan.restype = C.c_int
        L.orc_path_dkan.argtypes = [C.c_double, dp, dp, C.c_int]
        u32p = C.POINTER(C.c_uint32)
        L.orc_philox4x32_10.argtypes = [u32p, u32p, u32p]
        L.orc_normal_pair.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
        L.orc_normals.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Oracle:
    """One controller instance of the restated reference (same method names as the
    reference classes: sampling / predict_States / calc_Weights / determine_OptimalSolution)."""

    def __init__(self, model, K, H, sigma, lam, v_ref, u_min, u_max, path_weight=1.0, v_weight=1.0, zmp_weight=1.0,
                 roll_v_weight=1.0, back_weight=1.0, yaw_weight=1.0, roll_off=False, steer_off=False):
        m = MODELS[model] if isinstance(model, str) else int(model)
        self.model, self.K, self.H, self.udim = m, int(K), int(H), UDIM[m]
        cfg = OrcConfig()
        cfg.model, cfg.K, cfg.H = m, int(K), int(H)
        cfg.roll_off, cfg.steer_off = int(roll_off), int(steer_off)
        cfg.sigma, cfg.lam, cfg.v_ref = sigma, lam, v_ref
        for d in range(self.udim):
            cfg.u_min[d], cfg.u_max[d] = u_min[d], u_max[d]
        cfg.path_weight, cfg.v_weight, cfg.zmp_weight = path_weight, v_weight, zmp_weight
        cfg.roll_v_weight, cfg.back_weight, cfg.yaw_weight = roll_v_weight, back_weight, yaw_weight
        self._h = C.c_void_p(lib().orc_create(C.byref(cfg)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_destroy(self._h)
            self._h = None

    def set_by_value(self, on=True):
        """calc_Cost / calc_MinDistance take their arguments by value, as the reference's signatures do (dd.h:130,140;
        SURVEY.md Q16): the same values at the reference's memory behaviour (bench.py's reference-shaped CPU baseline)."""
        lib().orc_set_by_value(self._h, 1 if on else 0)

    # --- nominal (optimal_solution) ---
    def set_nominal(self, u):
        u = _f64(u).reshape(self.H - 1, self.udim)
        lib().orc_set_nominal(self._h, _dp(u))

    def get_nominal(self):
        u = np.empty((self.H - 1, self.udim))
        lib().orc_get_nominal(self._h, _dp(u))
        return u

    # --- the four hot methods (reference call order dd:352-358) ---
    def sampling(self, seed, rng="mt19937", iteration=0, k_offset=0):
        if rng == "mt19937":
            lib().orc_sampling_mt19937(self._h, int(seed) & 0xFFFFFFFF)
        else:
            lib().orc_sampling_philox(self._h, int(seed), int(iteration), int(k_offset))

    def set_controls(self, u):
        u = _f64(u).reshape(self.K, self.H - 1, self.udim)
        lib().orc_set_controls(self._h, _dp(u))

    def get_controls(self):
        u = np.empty((self.K, self.H - 1, self.udim))
        lib().orc_get_controls(self._h, _dp(u))
        return u

    def predict_States(self, x0, dt):
        x = np.zeros(5)
        x[:len(x0)] = x0
        lib().orc_predict_states(self._h, _dp(x), float(dt))

    def calc_Weights(self, x_ref, y_ref, yaw_ref0=0.0):
        xr, yr = _f64(x_ref), _f64(y_ref)
        assert xr.shape == (self.H,) and yr.shape == (self.H,)
        lib().orc_calc_weights(self._h, _dp(xr), _dp(yr), float(yaw_ref0))

    def determine_OptimalSolution(self):
        lib().orc_determine_optimal(self._h)
        return self.get_nominal()

    def iterate(self, x0, dt, x_ref, y_ref, yaw_ref0, seed, rng="mt19937", iteration=0, k_offset=0):
        x = np.zeros(5)
        x[:len(x0)] = x0
        xr, yr = _f64(x_ref), _f64(y_ref)
        out = np.empty((self.H - 1, self.udim))
        mode = {"mt19937": 0, "philox": 1, "inject": 2}[rng]
        lib().orc_iterate(self._h, mode, int(seed), int(iteration), int(k_offset), _dp(x), float(dt), _dp(xr), _dp(yr),
                          float(yaw_ref0), _dp(out))
        return out

    # --- read-back ---
    def costs(self):
        out = np.empty(self.K)
        lib().orc_get_costs(self._h, _dp(out))
        return out

    def weights(self):
        out = np.empty(self.K)
        lib().orc_get_weights(self._h, _dp(out))
        return out

    def sum_w(self):
        return lib().orc_get_sum_w(self._h)

    def states(self, which):
        idx = {"x": 0, "y": 1, "yaw": 2, "roll": 3, "pitch": 4, "zmp_x": 5, "zmp_y": 6}[which]
        n = self.H if idx < 5 else self.H - 2
        out = np.empty((self.K, n))
        lib().orc_get_states(self._h, idx, _dp(out))
        return out


class FbEstimator:
    """Restatement of the full-body state estimator: imuCallback fb:199-237, wrenchCallback fb:115-156,
    calc_true_ZMP fb:569-596, get_CurrentState fb:528-567."""

    def __init__(self):
        self._h = C.c_void_p(lib().orc_fbest_create())

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_fbest_destroy(self._h)
            self._h = None

    def imu(self, quat_xyzw, ang_vel, lin_acc, basis=None):
        b = _f64(np.eye(3) if basis is None else basis).reshape(9)
        lib().orc_fbest_imu(self._h, _dp(_f64(quat_xyzw)), _dp(_f64(ang_vel)), _dp(_f64(lin_acc)), _dp(b))

    def wrench(self, sensor, force, basis=None):
        b = _f64(np.eye(3) if basis is None else basis).reshape(9)
        lib().orc_fbest_wrench(self._h, int(sensor), _dp(_f64(force)), _dp(b))

    def update(self, x, y, yaw, dt):
        """calc_true_ZMP() then get_CurrentState() as run() does (fb:623-625); returns 0 when the denominator was too small."""
        ok = lib().orc_fbest_true_zmp(self._h)
        lib().orc_fbest_state(self._h, float(x), float(y), float(yaw), float(dt))
        return ok

    def read(self):
        out = np.zeros(16)
        lib().orc_fbest_read(self._h, _dp(out))
        return out


def calc_ref_path(path_x, path_y, cur_x, cur_y, v_ref, dt, resolution, H):
    px, py = _f64(path_x), _f64(path_y)
    xr, yr, yaw = np.zeros(H), np.zeros(H), np.zeros(H)
    idx = lib().orc_calc_ref_path(_dp(px), _dp(py), len(px), cur_x, cur_y, v_ref, dt, resolution, H, _dp(xr), _dp(yr),
                                  _dp(yaw))
    return idx, xr, yr, yaw


def path_cosine(A=(0.0, 0.0, 0.0), omega=(0.0, 0.0, 0.0), delta=(1.57, 1.57, 1.57), resolution=0.1, course_length=10.0,
                init_x=0.0, init_y=0.0):
    cap = int(course_length / resolution) + 16
    px, py = np.zeros(cap), np.zeros(cap)
    n = lib().orc_path_cosine(A[0], A[1], A[2], omega[0], omega[1], omega[2], delta[0], delta[1], delta[2], resolution,
                              course_length, init_x, init_y, _dp(px), _dp(py), cap)
    return px[:n].copy(), py[:n].copy()


def path_dkan(resolution=0.1):
    cap = 4096
    px, py = np.zeros(cap), np.zeros(cap)
    n = lib().orc_path_dkan(resolution, _dp(px), _dp(py), cap)
    return px[:n].copy(), py[:n].copy()


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(v) for v in o]


def normal_pair(a, b):
    z = (C.c_float * 2)()
    lib().orc_normal_pair(a, b, z)
    return np.float32(z[0]), np.float32(z[1])


def normals(seed, iteration, k0, nk, n_per):
    out = np.empty((nk, n_per), dtype=np.float32)
    lib().orc_normals(seed, iteration, k0, nk, n_per, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out
