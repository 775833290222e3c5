"""Python handle on the ROS-free mirror of the reference controller classes (include/ccv_mppi_node.hpp:
DiffDriveMPPI / SteeringDiffDriveMPPI / FullBodyMPPI with the reference's parameter names and defaults)."""
import ctypes as C

import numpy as np

from . import capi
from .configs import MODEL_IDS, NSTATE, UDIM
from .controller import MPPIError


class ControllerNode:
    """One controller node: set the path and the current state, then run_once(dt) = one pass of run() (dd:346-361)."""

    def __init__(self, model, params=None, device=0, seed=42, fused=True, device_prologue=False):
        self.lib = capi.load()
        self.model = model
        self.udim, self.nstate = UDIM[model], NSTATE[model]
        params = dict(params or {})
        defaults = {"diff_drive": (15, 1000), "steering_diff_drive": (15, 10000), "full_body": (15, 10000)}[model]
        self.H = int(params.get("horizon", defaults[0]))
        names = (C.c_char_p * len(params))(*[k.encode() for k in params])
        values = (C.c_double * len(params))(*[float(v) for v in params.values()])
        self._h = capi._H()
        rc = self.lib.ccv_mppi_node_create(MODEL_IDS[model], names, values, len(params), device, C.byref(self._h))
        if rc != capi.OK:
            self._h = capi._H()
            raise MPPIError(rc, "ccv_mppi_node_create failed -- no usable HIP device? (there is no CPU fallback)")
        self.lib.ccv_mppi_node_set_seed(self._h, int(seed))
        self.lib.ccv_mppi_node_set_fused(self._h, 1 if fused else 0)
        self.lib.ccv_mppi_node_set_device_prologue(self._h, 1 if device_prologue else 0)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.ccv_mppi_node_destroy(self._h)
            self._h = capi._H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_path(self, px, py):
        px, py = capi.as_f64(px), capi.as_f64(py)
        rc = self.lib.ccv_mppi_node_set_path(self._h, capi.dptr(px), capi.dptr(py), len(px))
        if rc != capi.OK:
            raise MPPIError(rc, "set_path")

    def set_state(self, state):
        s = np.zeros(5)
        s[:len(state)] = state
        self.lib.ccv_mppi_node_set_state(self._h, capi.dptr(s))

    def run_once(self, dt):
        """Returns None while no path was received, else dict(cmd_vel=(v, w), cmd_pos=(steer_l, steer_r, fore, rear, roll))."""
        cmd = np.zeros(7)
        rc = self.lib.ccv_mppi_node_run_once(self._h, float(dt), capi.dptr(cmd))
        if rc < 0:
            raise MPPIError(rc, "run_once")
        if rc == 0:
            return None
        return {"cmd_vel": (cmd[0], cmd[1]), "cmd_pos": tuple(cmd[2:7])}

    def optimal_solution(self):
        u = np.zeros((self.H - 1, self.udim))
        self.lib.ccv_mppi_node_get_optimal(self._h, capi.dptr(u))
        return u

    def ref_path(self):
        out = np.zeros((self.H, 3))
        self.lib.ccv_mppi_node_get_ref_path(self._h, capi.dptr(out))
        return out

    def optimal_path(self):
        out = np.zeros((self.H - 1, 3))
        self.lib.ccv_mppi_node_get_optimal_path(self._h, capi.dptr(out))
        return out

    # ---- full-body state estimator (fb:115-156,188-237,528-596); full_body nodes only ----
    def fb_imu(self, quat_xyzw, angular_velocity, linear_acceleration, basis=None):
        b = None if basis is None else capi.as_f64(basis).reshape(9)
        rc = self.lib.ccv_mppi_node_fb_imu(self._h, capi.dptr(capi.as_f64(quat_xyzw)), capi.dptr(capi.as_f64(angular_velocity)),
                                           capi.dptr(capi.as_f64(linear_acceleration)), None if b is None else capi.dptr(b))
        if rc != capi.OK:
            raise MPPIError(rc, "fb_imu")

    def fb_wrench(self, sensor, force, basis=None):
        b = None if basis is None else capi.as_f64(basis).reshape(9)
        rc = self.lib.ccv_mppi_node_fb_wrench(self._h, int(sensor), capi.dptr(capi.as_f64(force)), None if b is None else capi.dptr(b))
        if rc != capi.OK:
            raise MPPIError(rc, "fb_wrench")

    def fb_pose(self, x, y, yaw):
        rc = self.lib.ccv_mppi_node_fb_pose(self._h, float(x), float(y), float(yaw))
        if rc != capi.OK:
            raise MPPIError(rc, "fb_pose")

    def fb_read(self):
        """current_state_ (5), zmp_x, zmp_y, true_ZMP (3), imu roll / pitch / yaw, accel x / y / z"""
        out = np.zeros(16)
        rc = self.lib.ccv_mppi_node_fb_read(self._h, capi.dptr(out))
        if rc != capi.OK:
            raise MPPIError(rc, "fb_read")
        return out


class FullBodyStateEstimator:
    """The estimator of FullBodyMPPI on its own (host only, no device): include/ccv_mppi_node.hpp."""

    def __init__(self):
        self.lib = capi.load()
        self._h = capi._H()
        rc = self.lib.ccv_mppi_fb_estimator_create(C.byref(self._h))
        if rc != capi.OK:
            raise MPPIError(rc, "ccv_mppi_fb_estimator_create")

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.ccv_mppi_fb_estimator_destroy(self._h)
            self._h = capi._H()

    def imu(self, quat_xyzw, angular_velocity, linear_acceleration, basis=None):
        b = None if basis is None else capi.as_f64(basis).reshape(9)
        rc = self.lib.ccv_mppi_fb_estimator_imu(self._h, capi.dptr(capi.as_f64(quat_xyzw)), capi.dptr(capi.as_f64(angular_velocity)),
                                                capi.dptr(capi.as_f64(linear_acceleration)), None if b is None else capi.dptr(b))
        if rc != capi.OK:
            raise MPPIError(rc, "estimator imu")

    def wrench(self, sensor, force, basis=None):
        b = None if basis is None else capi.as_f64(basis).reshape(9)
        rc = self.lib.ccv_mppi_fb_estimator_wrench(self._h, int(sensor), capi.dptr(capi.as_f64(force)), None if b is None else capi.dptr(b))
        if rc != capi.OK:
            raise MPPIError(rc, "estimator wrench")

    def update(self, x, y, yaw, dt):
        rc = self.lib.ccv_mppi_fb_estimator_update(self._h, float(x), float(y), float(yaw), float(dt))
        if rc < 0:
            raise MPPIError(rc, "estimator update")
        return rc

    def read(self):
        out = np.zeros(16)
        self.lib.ccv_mppi_fb_estimator_read(self._h, capi.dptr(out))
        return out
