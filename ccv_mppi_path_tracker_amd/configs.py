"""Parameter sets of the three reference controllers and the BASELINE.json workloads C1-C5.

Defaults are the in-code defaults of the reference constructors; launch overrides are the values the
reference's launch files set:
  diff_drive            src/diff_drive_mppi.cpp:17-34          launch/diff_drive_mppi.launch:6-9
  steering_diff_drive   src/steering_diff_drive_mppi.cpp:18-36 launch/steering_diff_drive_mppi.launch:7-10
  full_body             src/full_body_mppi.cpp:8-42            launch/full_body_mppi.launch:7-15
"""
import math
from dataclasses import dataclass, field, replace
from typing import Tuple

DEG = math.pi / 180.0
MODEL_IDS = {"diff_drive": 0, "steering_diff_drive": 1, "full_body": 2}
UDIM = {"diff_drive": 2, "steering_diff_drive": 3, "full_body": 5}
NSTATE = {"diff_drive": 3, "steering_diff_drive": 3, "full_body": 5}
CONTROL_NAMES = {
    "diff_drive": ("v", "w"),
    "steering_diff_drive": ("v", "w", "steer"),
    "full_body": ("v", "w", "direction", "roll_v", "pitch_v"),
}


@dataclass(frozen=True)
class MPPIParams:
    model: str
    num_samples: int
    horizon: int
    control_noise: float = 0.5
    lam: float = 1.0
    v_ref: float = 0.8
    u_min: Tuple[float, ...] = ()
    u_max: Tuple[float, ...] = ()
    path_weight: float = 1.0
    v_weight: float = 1.0
    zmp_weight: float = 1.0
    roll_v_weight: float = 1.0
    back_weight: float = 1.0
    yaw_weight: float = 1.0
    roll_off: bool = False
    steer_off: bool = False
    dt: float = 0.1
    resolution: float = 0.1

    @property
    def udim(self):
        return UDIM[self.model]

    @property
    def nstate(self):
        return NSTATE[self.model]

    def with_(self, **kw):
        return replace(self, **kw)


def diff_drive_defaults(num_samples=1000, horizon=15):
    """src/diff_drive_mppi.cpp:17-34"""
    return MPPIParams("diff_drive", num_samples, horizon, v_ref=0.8, u_min=(-1.2, -2.0), u_max=(1.2, 2.0))


def steering_defaults(num_samples=10000, horizon=15):
    """src/steering_diff_drive_mppi.cpp:18-36"""
    return MPPIParams("steering_diff_drive", num_samples, horizon, v_ref=0.8,
                      u_min=(-1.2, -1.0, -30.0 * DEG), u_max=(1.2, 1.0, 30.0 * DEG))


def full_body_defaults(num_samples=10000, horizon=15):
    """src/full_body_mppi.cpp:8-42"""
    return MPPIParams("full_body", num_samples, horizon, v_ref=1.2,
                      u_min=(-3.0, -1.0, -30.0 * DEG, -30.0 * DEG, -15.0 * DEG),
                      u_max=(1.2, 1.0, 30.0 * DEG, 30.0 * DEG, 15.0 * DEG))


@dataclass(frozen=True)
class Workload:
    name: str
    params: MPPIParams
    path: str                      # "straight" | "sinusoid" | "dkan"
    description: str = ""
    x0: Tuple[float, ...] = field(default=(0.0, 0.0, 0.0))


def workload(name, num_samples=None, horizon=None):
    """BASELINE.json configs."""
    if name == "C1":   # dd CPU plumbing case: code defaults, straight path
        p = diff_drive_defaults(256, 30)
        w = Workload("C1", p, "straight", "diff_drive K=256 T=30 straight")
    elif name in ("C2", "C5"):   # dd, launch values, sinusoid (C5 = C2 inputs, K sharded over 8 GPUs)
        p = diff_drive_defaults(65536 if name == "C2" else 524288, 50).with_(
            path_weight=10.0, v_ref=1.2, u_max=(2.0, 2.0))
        w = Workload(name, p, "sinusoid", "diff_drive K=%d T=50 sinusoid" % p.num_samples)
    elif name == "C3":
        p = steering_defaults(65536, 50).with_(path_weight=10.0, v_ref=1.2, u_max=(2.0, 1.0, 30.0 * DEG))
        w = Workload("C3", p, "sinusoid", "steering_diff_drive K=65536 T=50 sinusoid")
    elif name == "C4":   # launch weights with roll_off=false so the ZMP term is live
        p = full_body_defaults(131072, 80).with_(
            v_ref=2.0, u_max=(2.0, 1.0, 30.0 * DEG, 30.0 * DEG, 15.0 * DEG), path_weight=10.0, v_weight=1.0,
            zmp_weight=10.0, roll_v_weight=0.5, back_weight=1.0, yaw_weight=2.0, roll_off=False)
        w = Workload("C4", p, "dkan", "full_body K=131072 T=80 dkan", x0=(0.0, 0.0, 0.0, 0.0, 0.0))
    else:
        raise KeyError(name)
    if num_samples is not None or horizon is not None:
        p = w.params.with_(num_samples=num_samples or w.params.num_samples, horizon=horizon or w.params.horizon)
        w = replace(w, params=p)
    return w
