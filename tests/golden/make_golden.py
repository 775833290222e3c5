#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (oracle/mppi_oracle.cpp).

The reference ships no tests or fixtures and cannot be built in this image (DESIGN.md
"Oracle"), so these vectors pin the ORACLE (regression + cross-machine libm check), not the reference:
parity stays "unpinned" in the sense of DESIGN.md.  Each file holds three consecutive closed-loop iterations
(mt19937 mode, the reference's own generator) of one small case: inputs, the new optimal controls, sum of
weights, all K costs and weights, and the controls/states of a few samples.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
N_ITERS = 3
SEED0 = 42


def run_case(name, p, path_kind):
    o = helpers.oracle_for(p)
    path = helpers.oracle_path(path_kind)
    state = np.zeros(p.nstate)
    state[0], state[1] = path[0][0], path[1][0]
    rec = {"path_x": path[0], "path_y": path[1]}
    for it in range(N_ITERS):
        xr, yr, yaw = helpers.oracle_window(p, path, state)
        u_in = o.get_nominal()
        u_out = o.iterate(state, p.dt, xr, yr, yaw[0], seed=SEED0 + it, rng="mt19937")
        pre = "it%d_" % it
        rec[pre + "x0"] = state.copy()
        rec[pre + "x_ref"], rec[pre + "y_ref"], rec[pre + "yaw_ref0"] = xr, yr, np.float64(yaw[0])
        rec[pre + "seed"] = np.int64(SEED0 + it)
        rec[pre + "u_in"], rec[pre + "u_out"] = u_in, u_out
        rec[pre + "sum_w"] = np.float64(o.sum_w())
        rec[pre + "costs"], rec[pre + "weights"] = o.costs(), o.weights()
        ctrl = o.get_controls()
        sel = [0, 1, p.num_samples - 1]
        rec[pre + "controls_sel"] = ctrl[sel]
        rec[pre + "x_sel"] = o.states("x")[sel]
        rec[pre + "y_sel"] = o.states("y")[sel]
        if p.model == "full_body":
            rec[pre + "zmp_y_sel"] = o.states("zmp_y")[sel]
        state = helpers.plant(p.model, state, u_out[0], p.dt)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    return rec


if __name__ == "__main__":
    for name, p, kind in helpers.small_cases():
        r = run_case(name, p, kind)
        print(name, "u_out[0] =", r["it2_u_out"][0], "sum_w =", r["it2_sum_w"])
