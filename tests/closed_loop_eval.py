#!/usr/bin/env python3
"""Closed-loop tracking error of the GPU controller next to the CPU oracle: the reference's own notion of
"works" -- src/record_state.py:118-139 logs the pose against the path, src/calc_e_rmse.py:30-49 reports the maximum and
the RMS of the distance from each logged pose to the nearest path point.  Here both controllers (same parameters, same
Philox noise) drive the same kinematic plant (the Euler model the controller itself assumes) along the sinusoid of
launch/diff_drive_mppi.launch and the dkan path for `steps` control periods.

  python tests/closed_loop_eval.py [--steps 150] [--samples 2048]        (needs a GPU; the oracle runs on the host)

Lives under tests/ because it runs the oracle (test infrastructure) as the checker; not collected by pytest.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))   # helpers.py
import ccv_mppi_path_tracker_amd as amd  # noqa: E402
from ccv_mppi_path_tracker_amd import configs  # noqa: E402
import helpers  # noqa: E402  (tests/helpers.py: oracle construction, test infrastructure)


def tracking_errors(traj, px, py):
    """calc_e_rmse.py:30-49"""
    d = np.sqrt((traj[:, 0:1] - px[None, :]) ** 2 + (traj[:, 1:2] - py[None, :]) ** 2).min(axis=1)
    return float(d.max()), float(np.sqrt(np.mean(d * d)))


def run(wl, path_kind, steps, samples, seed=2025):
    w = configs.workload(wl, num_samples=samples)
    p = w.params
    px, py = amd.make_path(path_kind)
    s_g = np.zeros(p.nstate)
    s_g[0], s_g[1] = px[0], py[0]
    s_g[2] = np.arctan2(py[1] - py[0], px[1] - px[0])
    s_o = s_g.copy()
    g, o = amd.MPPIController(p), helpers.oracle_for(p)
    tg, to = [s_g.copy()], [s_o.copy()]
    for it in range(steps):
        _, xr, yr, yaw = amd.calc_ref_path(px, py, s_g[0], s_g[1], p.v_ref, p.dt, p.resolution, p.horizon)
        u_g = g.iterate(s_g, p.dt, xr, yr, yaw[0], seed, it, want_stats=False)
        xo, yo, yawo = helpers.oracle_window(p, (px, py), s_o)
        u_o = o.iterate(s_o, p.dt, xo, yo, yawo[0], seed=seed, rng="philox", iteration=it)
        s_g = amd.plant_step(p.model, s_g, u_g[0], p.dt)
        s_o = helpers.plant(p.model, s_o, u_o[0], p.dt)
        tg.append(s_g.copy())
        to.append(s_o.copy())
    tg, to = np.array(tg), np.array(to)
    mg, rg = tracking_errors(tg, px, py)
    mo, ro = tracking_errors(to, px, py)
    apart = float(np.max(np.abs(tg[:, :3] - to[:, :3])))
    return dict(workload=wl, path=path_kind, steps=steps, K=samples, gpu_max=mg, gpu_rmse=rg, cpu_max=mo, cpu_rmse=ro,
                max_pose_difference=apart, travelled=float(np.hypot(*(tg[-1, :2] - tg[0, :2]))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--samples", type=int, default=2048)
    ap.add_argument("--seeds", type=int, default=3)
    a = ap.parse_args()
    # The closed loop amplifies rounding differences (x1.4 per control period with these parameters): the two runs of one
    # seed stay together for ~60 periods (tests/test_gpu_parity.py::test_closed_loop_matches_oracle_and_tracks) and are
    # independent realisations afterwards, so the comparison is between the error statistics over several seeds.
    print("%-4s %-9s %6s %6s %5s | %-27s | %-27s | %s" % ("wl", "path", "steps", "K", "seeds", "GPU max / RMSE [m] (mean)",
                                                            "CPU max / RMSE [m] (mean)", "poses apart after 40 periods"))
    for wl, kind in (("C2", "sinusoid"), ("C2", "dkan"), ("C3", "sinusoid"), ("C3", "dkan"), ("C4", "sinusoid"), ("C4", "dkan")):
        rs = [run(wl, kind, a.steps, a.samples, seed=2025 + i) for i in range(a.seeds)]
        early = max(run(wl, kind, 40, a.samples, seed=2025)["max_pose_difference"] for _ in range(1))
        m = lambda k: float(np.mean([r[k] for r in rs]))   # noqa: E731
        print("%-4s %-9s %6d %6d %5d | %12.4f / %12.4f | %12.4f / %12.4f | %.2e" % (wl, kind, a.steps, a.samples, a.seeds,
              m("gpu_max"), m("gpu_rmse"), m("cpu_max"), m("cpu_rmse"), early))


if __name__ == "__main__":
    main()
