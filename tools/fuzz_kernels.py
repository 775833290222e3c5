#!/usr/bin/env python3
"""Diagnostic (GPU box): random horizons and sample counts -- diff drive and steering: the four-wave kernel against the
three-wave kernel (every bit) and against the one-wave kernel (samples and states bit for bit, the rest to rounding); full
body: the default kernel (four-wave up to one block per CU) against the two-wave and the one-wave kernel (samples and states
bit for bit, the rest to rounding); two iterations each, one of them from a warm start with a NaN in it now and then.
   python tools/fuzz_kernels.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ccv_mppi_path_tracker_amd as amd  # noqa: E402
from ccv_mppi_path_tracker_amd import configs  # noqa: E402
import helpers  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def make(p, kern):
    os.environ.pop("CCV_MPPI_KERNEL", None)
    if kern:
        os.environ["CCV_MPPI_KERNEL"] = kern
    g = amd.MPPIController(p)
    os.environ.pop("CCV_MPPI_KERNEL", None)
    return g


bad = 0
for case in range(n_cases):
    wl = ["C2", "C3", "C4"][int(rng.integers(3))]
    H = int(rng.choice([3, 4, 8, 9, 10, 16, 17, 24, 25, 26, 33, 40, 41, 49, 50, 51, 57, 64, 65, 80, 100, 127, 128]))
    K = int(rng.choice([1, 63, 64, 65, 200, 777, 1024, 4097, 16384]))
    w = configs.workload(wl)
    p = w.params.with_(num_samples=K, horizon=H, control_noise=float(rng.choice([0.1, 0.5, 1.5])))
    path = helpers.oracle_path(w.path)
    state = np.array(path[0][:1].tolist() + path[1][:1].tolist() + [0.0]) if False else None
    from test_gpu_parity import start_state  # noqa: E402
    state = start_state(p, path)
    xr, yr, yaw = helpers.oracle_window(p, path, state)
    fb = wl == "C4"
    gs = {k: make(p, k) for k in ((None, "pc", "solo") if fb else (None, "r3", "solo"))}
    if fb:
        gs["r3"] = gs.pop("pc")   # (the cross-check kernel of this model)
    nan_case = rng.random() < 0.2
    if nan_case:
        nom = np.zeros((H - 1, p.udim))
        nom[int(rng.integers(H - 1)), int(rng.integers(p.udim))] = np.nan
        for g in gs.values():
            g.set_nominal(nom)
    ok = True
    for it in range(2):
        u = {k: g.iterate(state, p.dt, xr, yr, yaw[0], 3 + case, it, want_stats=False) for k, g in gs.items()}
        if not fb:
            ok &= np.array_equal(u[None], u["r3"], equal_nan=True)
            ok &= np.array_equal(gs[None].read_costs(), gs["r3"].read_costs(), equal_nan=True)
            ok &= np.array_equal(gs[None].read_controls(), gs["r3"].read_controls(), equal_nan=True)
            ok &= np.array_equal(gs[None].read_candidates(), gs["r3"].read_candidates(), equal_nan=True)
        elif it == 0 and not nan_case:
            ok &= np.array_equal(gs[None].read_controls(), gs["r3"].read_controls(), equal_nan=True)
            ok &= np.array_equal(gs[None].read_candidates(), gs["r3"].read_candidates(), equal_nan=True)
            ok &= np.allclose(gs[None].read_costs(), gs["r3"].read_costs(), rtol=1e-11, equal_nan=True)
            ok &= np.allclose(u[None], u["r3"], rtol=1e-8, atol=1e-11, equal_nan=True)
        ok &= np.array_equal(gs[None].read_controls(), gs["solo"].read_controls(), equal_nan=True) if it == 0 else True
        if not nan_case:
            ok &= np.allclose(u[None], u["solo"], rtol=1e-8, atol=1e-11, equal_nan=True)
            ok &= bool(np.all(np.isfinite(u[None]))) or float(np.nansum(gs[None].read_weights())) == 0.0
    print("%s H=%3d K=%5d sigma=%.1f nan=%d  %s" % (wl, H, K, p.control_noise, nan_case, "ok" if ok else "MISMATCH"), flush=True)
    bad += not ok
    for g in gs.values():
        g.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
