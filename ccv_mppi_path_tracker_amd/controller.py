"""Thin Python host over the C ABI.  Method names follow the reference controller classes
(sampling / predict_States / calc_Weights / determine_OptimalSolution, src/diff_drive_mppi.cpp:81,111,212,225)
so the parity tests read like calls into the reference.  All compute happens in libccv_mppi_hip.so.
"""
import ctypes as C

import numpy as np

from . import capi
from .configs import MODEL_IDS, MPPIParams


class MPPIError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ccv_mppi error %d: %s" % (code, msg))
        self.code = code


def make_config(p: MPPIParams, device=0, sample_offset=0, min_shift=False, no_state_store=False, num_samples=None):
    cfg = capi.Config()
    cfg.abi_version = capi.ABI_VERSION
    cfg.model = MODEL_IDS[p.model]
    cfg.num_samples = int(num_samples if num_samples is not None else p.num_samples)
    cfg.horizon = int(p.horizon)
    cfg.sample_offset = int(sample_offset)
    cfg.device = int(device)
    cfg.flags = ((capi.FLAG_ROLL_OFF if p.roll_off else 0) | (capi.FLAG_STEER_OFF if p.steer_off else 0) |
                 (capi.FLAG_MIN_SHIFT if min_shift else 0) | (capi.FLAG_NO_STATE_STORE if no_state_store else 0))
    cfg.control_noise, cfg.lam, cfg.v_ref = p.control_noise, p.lam, p.v_ref
    for d in range(p.udim):
        cfg.u_min[d], cfg.u_max[d] = p.u_min[d], p.u_max[d]
    cfg.path_weight, cfg.v_weight, cfg.zmp_weight = p.path_weight, p.v_weight, p.zmp_weight
    cfg.roll_v_weight, cfg.back_weight, cfg.yaw_weight = p.roll_v_weight, p.back_weight, p.yaw_weight
    return cfg


class MPPIController:
    """One handle of the C ABI = one reference controller instance (K samples on one device)."""

    def __init__(self, params: MPPIParams, device=0, sample_offset=0, num_samples=None, min_shift=False,
                 no_state_store=False):
        self.lib = capi.load()
        self.params = params
        self.K = int(num_samples if num_samples is not None else params.num_samples)
        self.H = params.horizon
        self.udim = params.udim
        self.nstate = params.nstate
        self._h = capi._H()
        cfg = make_config(params, device, sample_offset, min_shift, no_state_store, self.K)
        rc = self.lib.ccv_mppi_create(C.byref(cfg), C.byref(self._h))
        if rc != capi.OK:
            self._h = capi._H()
            raise MPPIError(rc, "ccv_mppi_create failed (no usable MI355X/HIP device?) -- there is no CPU fallback")

    # ---- plumbing ----
    def _check(self, rc):
        if rc != capi.OK:
            msg = self.lib.ccv_mppi_last_error(self._h)
            raise MPPIError(rc, msg.decode() if msg else "")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.ccv_mppi_destroy(self._h)
            self._h = capi._H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _x0(self, x0):
        x = np.zeros(5)
        x[:len(x0)] = x0
        return x

    def set_stream(self, stream_ptr):
        self._check(self.lib.ccv_mppi_set_stream(self._h, C.c_void_p(stream_ptr)))

    def synchronize(self):
        self._check(self.lib.ccv_mppi_synchronize(self._h))

    # ---- optimal_solution (warm start) ----
    def set_nominal(self, u):
        u = capi.as_f64(u, (self.H - 1, self.udim))
        self._check(self.lib.ccv_mppi_set_nominal(self._h, capi.dptr(u)))

    def get_nominal(self):
        u = np.empty((self.H - 1, self.udim))
        self._check(self.lib.ccv_mppi_get_nominal(self._h, capi.dptr(u)))
        return u

    # ---- whole iteration ----
    def iterate(self, x0, dt, x_ref, y_ref, yaw_ref0, seed, iteration, want_stats=True):
        x = self._x0(x0)
        xr, yr = capi.as_f64(x_ref, (self.H,)), capi.as_f64(y_ref, (self.H,))
        u = np.empty((self.H - 1, self.udim))
        st = capi.Stats()
        self._check(self.lib.ccv_mppi_iterate(self._h, capi.dptr(x), float(dt), capi.dptr(xr), capi.dptr(yr),
                                              float(yaw_ref0), int(seed), int(iteration), capi.dptr(u),
                                              C.byref(st) if want_stats else None))
        return (u, st) if want_stats else u

    def iterate_enqueue(self, x0, dt, x_ref, y_ref, yaw_ref0, seed, iteration):
        x = self._x0(x0)
        xr, yr = capi.as_f64(x_ref, (self.H,)), capi.as_f64(y_ref, (self.H,))
        self._check(self.lib.ccv_mppi_iterate_enqueue(self._h, capi.dptr(x), float(dt), capi.dptr(xr), capi.dptr(yr),
                                                      float(yaw_ref0), int(seed), int(iteration)))

    # ---- the four reference methods, stage-wise ----
    def sampling(self, seed, iteration):
        self._check(self.lib.ccv_mppi_sample(self._h, int(seed), int(iteration)))

    def inject_controls(self, u_samples):
        u = capi.as_f64(u_samples, (self.K, self.H - 1, self.udim))
        self._check(self.lib.ccv_mppi_inject_controls(self._h, capi.dptr(u)))

    def predict_States(self, x0, dt):
        x = self._x0(x0)
        self._check(self.lib.ccv_mppi_rollout(self._h, capi.dptr(x), float(dt)))

    def calc_Weights(self, x_ref, y_ref, yaw_ref0=0.0):
        xr, yr = capi.as_f64(x_ref, (self.H,)), capi.as_f64(y_ref, (self.H,))
        self._check(self.lib.ccv_mppi_weights(self._h, capi.dptr(xr), capi.dptr(yr), float(yaw_ref0)))

    def determine_OptimalSolution(self, want_stats=False):
        u = np.empty((self.H - 1, self.udim))
        st = capi.Stats()
        self._check(self.lib.ccv_mppi_update(self._h, capi.dptr(u), C.byref(st)))
        return (u, st) if want_stats else u

    # ---- read-back ----
    def read_candidates(self, first=0, count=None, stride=1):
        count = self.K if count is None else count
        out = np.empty((count, self.H, 2))
        self._check(self.lib.ccv_mppi_read_candidates(self._h, first, count, stride, capi.dptr(out)))
        return out

    def read_top_candidates(self, count, with_paths=True):
        """The `count` highest-weight samples: (indices, unnormalised weights, rollouts [count][H][2] or None)."""
        idx = np.empty(count, dtype=np.int32)
        wts = np.empty(count)
        xy = np.empty((count, self.H, 2)) if with_paths else None
        self._check(self.lib.ccv_mppi_read_top_candidates(self._h, int(count), idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                                          capi.dptr(wts), capi.dptr(xy) if with_paths else None))
        return idx, wts, xy

    def read_costs(self, first=0, count=None):
        count = self.K - first if count is None else count
        out = np.empty(count)
        self._check(self.lib.ccv_mppi_read_costs(self._h, first, count, capi.dptr(out)))
        return out

    def read_weights(self, first=0, count=None):
        count = self.K - first if count is None else count
        out = np.empty(count)
        self._check(self.lib.ccv_mppi_read_weights(self._h, first, count, capi.dptr(out)))
        return out

    def read_controls(self, first=0, count=None):
        count = self.K - first if count is None else count
        out = np.empty((count, self.H - 1, self.udim))
        self._check(self.lib.ccv_mppi_read_controls(self._h, first, count, capi.dptr(out)))
        return out

    # ---- K sharded over devices ----
    def partials_size(self):
        return self.lib.ccv_mppi_partials_size(self._h)

    def iterate_partials_enqueue(self, x0, dt, x_ref, y_ref, yaw_ref0, seed, iteration, dev_ptr):
        x = self._x0(x0)
        xr, yr = capi.as_f64(x_ref, (self.H,)), capi.as_f64(y_ref, (self.H,))
        self._check(self.lib.ccv_mppi_iterate_partials_enqueue(
            self._h, capi.dptr(x), float(dt), capi.dptr(xr), capi.dptr(yr), float(yaw_ref0), int(seed),
            int(iteration), C.c_void_p(dev_ptr)))

    def apply_partials_enqueue(self, dev_ptr):
        self._check(self.lib.ccv_mppi_apply_partials_enqueue(self._h, C.c_void_p(dev_ptr)))

    # ---- direct exchange between the devices of one node (no collective call per iteration) ----
    def exchange_create(self, world, rank):
        """-> the IPC handle (bytes) of this device's box, to be gathered from all ranks."""
        buf = C.create_string_buffer(self.lib.ccv_mppi_exchange_handle_bytes())
        self._check(self.lib.ccv_mppi_exchange_create(self._h, int(world), int(rank), C.cast(buf, C.c_void_p)))
        return bytes(buf.raw)

    def exchange_connect(self, handles):
        blob = b"".join(handles)
        buf = C.create_string_buffer(blob, len(blob))
        self._check(self.lib.ccv_mppi_exchange_connect(self._h, C.cast(buf, C.c_void_p)))

    def exchange_info(self):
        """dict(world, rank, fine_grained, connected) of the direct exchange set up on this handle."""
        w, r, f, c = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self.lib.ccv_mppi_exchange_info(self._h, C.byref(w), C.byref(r), C.byref(f), C.byref(c)))
        return {"world": w.value, "rank": r.value, "fine_grained": bool(f.value), "connected": bool(c.value)}

    def iterate_exchange_enqueue(self, x0, dt, x_ref, y_ref, yaw_ref0, seed, iteration):
        x = self._x0(x0)
        xr, yr = capi.as_f64(x_ref, (self.H,)), capi.as_f64(y_ref, (self.H,))
        self._check(self.lib.ccv_mppi_iterate_exchange_enqueue(
            self._h, capi.dptr(x), float(dt), capi.dptr(xr), capi.dptr(yr), float(yaw_ref0), int(seed), int(iteration)))

    # ---- device-resident closed loop (get_CurrentIndex + calc_RefPath + plant on the device) ----
    def resident_set_path(self, path_x, path_y, resolution=None):
        px, py = capi.as_f64(path_x), capi.as_f64(path_y)
        res = self.params.resolution if resolution is None else resolution
        self._check(self.lib.ccv_mppi_resident_set_path(self._h, capi.dptr(px), capi.dptr(py), len(px), float(res)))

    def resident_set_pose(self, state):
        x = self._x0(state)
        self._check(self.lib.ccv_mppi_resident_set_pose(self._h, capi.dptr(x)))

    def resident_step_enqueue(self, dt, seed, iteration, advance=True):
        """One tick, no host data: (advance) pose += plant(u*[0]) -> window from the pose -> MPPI iteration."""
        self._check(self.lib.ccv_mppi_resident_step_enqueue(self._h, float(dt), int(seed), int(iteration),
                                                            1 if advance else 0))

    def resident_step_partials_enqueue(self, dt, seed, iteration, advance, dev_ptr):
        self._check(self.lib.ccv_mppi_resident_step_partials_enqueue(self._h, float(dt), int(seed), int(iteration),
                                                                     1 if advance else 0, C.c_void_p(dev_ptr)))

    def resident_step_exchange_enqueue(self, dt, seed, iteration, advance=True):
        self._check(self.lib.ccv_mppi_resident_step_exchange_enqueue(self._h, float(dt), int(seed), int(iteration),
                                                                     1 if advance else 0))

    def resident_read(self):
        """(state, current_index, x_ref, y_ref, yaw_ref0, steps) of the last tick; synchronises."""
        st = np.zeros(5)
        xr, yr = np.zeros(self.H), np.zeros(self.H)
        idx, yaw0, steps = C.c_int32(), C.c_double(), C.c_int64()
        self._check(self.lib.ccv_mppi_resident_read(self._h, capi.dptr(st), C.byref(idx), capi.dptr(xr), capi.dptr(yr),
                                                    C.byref(yaw0), C.byref(steps)))
        return st[:self.params.nstate].copy(), idx.value, xr, yr, yaw0.value, steps.value

    def resident_read_trace(self, max_rows=8192):
        """Poses of the last ticks, oldest first: rows (x, y, yaw, roll, pitch, current_index)."""
        rows = np.zeros((max_rows, 6))
        n = C.c_int32()
        self._check(self.lib.ccv_mppi_resident_read_trace(self._h, int(max_rows), capi.dptr(rows), C.byref(n)))
        return rows[:n.value].copy()

    # ---- measurement ----
    def timing_enable(self, on=True, every=1):
        self._check(self.lib.ccv_mppi_timing_enable(self._h, (max(1, int(every)) if on else 0)))

    def timing_read(self, reset=True):
        a, b, n = C.c_double(), C.c_double(), C.c_int64()
        self._check(self.lib.ccv_mppi_timing_read(self._h, C.byref(a), C.byref(b), C.byref(n), 1 if reset else 0))
        return a.value, b.value, n.value


# ---- host prologue (include/ccv_mppi_host.h) ----

def calc_ref_path(path_x, path_y, cur_x, cur_y, v_ref, dt, resolution, horizon):
    """get_CurrentIndex() + calc_RefPath() (src/diff_drive_mppi.cpp:126-140,156-181)."""
    lib = capi.load()
    px, py = capi.as_f64(path_x), capi.as_f64(path_y)
    xr, yr, yaw = np.zeros(horizon), np.zeros(horizon), np.zeros(horizon)
    idx = lib.ccv_mppi_calc_ref_path(capi.dptr(px), capi.dptr(py), len(px), cur_x, cur_y, v_ref, dt, resolution,
                                     horizon, capi.dptr(xr), capi.dptr(yr), capi.dptr(yaw))
    if idx < 0:
        raise MPPIError(idx, "ccv_mppi_calc_ref_path")
    return idx, xr, yr, yaw


def make_path(kind, resolution=0.1, length=None):
    """The synthetic reference paths of the BASELINE configs.  `length` overrides the course length of the
    cosine generator (launch value 10 m) for closed loops that must not run out of path."""
    lib = capi.load()
    cap = 4096 if length is None else int(length / resolution) + 16
    px, py = np.zeros(cap), np.zeros(cap)
    if kind == "dkan":   # dkan_path_creator.cpp
        n = lib.ccv_mppi_path_dkan(resolution, capi.dptr(px), capi.dptr(py), cap)
    else:
        if kind == "straight":   # creator defaults A=0, delta=1.57, length 10 (reference_path_creator.cpp:6-19)
            A, om, de, course = (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (1.57, 1.57, 1.57), 10.0
        elif kind == "sinusoid":   # launch/diff_drive_mppi.launch:15-26
            A, om, de, course = (1.0, 0.0, 0.0), (0.25, 0.0, 0.0), (0.0, 0.0, 0.0), 10.0
        else:
            raise KeyError(kind)
        A, om, de = capi.as_f64(A), capi.as_f64(om), capi.as_f64(de)
        n = lib.ccv_mppi_path_cosine(capi.dptr(A), capi.dptr(om), capi.dptr(de), resolution,
                                     course if length is None else float(length), 0.0, 0.0,
                                     capi.dptr(px), capi.dptr(py), cap)
    if n < 0:
        raise MPPIError(n, "path generator")
    return px[:n].copy(), py[:n].copy()


def plant_step(model, state, u, dt):
    lib = capi.load()
    s = capi.as_f64(state).copy()
    s5 = np.zeros(5)
    s5[:len(s)] = s
    uu = np.zeros(5)
    uu[:len(u)] = u
    rc = lib.ccv_mppi_plant_step(MODEL_IDS[model], capi.dptr(s5), capi.dptr(uu), float(dt))
    if rc != capi.OK:
        raise MPPIError(rc, "ccv_mppi_plant_step")
    return s5[:len(s)]
