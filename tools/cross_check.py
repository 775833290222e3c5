#!/usr/bin/env python3
"""Diagnostic (GPU box): two kernel variants (CCV_MPPI_KERNEL values, '' = default) over n iterations of the scripted loop --
are u*, costs, weights, controls and states the same bits?   python tools/cross_check.py WORKLOAD K n variantA variantB"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ccv_mppi_path_tracker_amd as amd  # noqa: E402
from ccv_mppi_path_tracker_amd import configs  # noqa: E402
import bench  # noqa: E402

wl, K, n, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
w = configs.workload(wl, num_samples=K)
p = w.params
inputs = bench.script_inputs(amd, w, 64)


def make(v):
    os.environ.pop("CCV_MPPI_KERNEL", None)
    if v not in ("", "default"):
        os.environ["CCV_MPPI_KERNEL"] = v
    return amd.MPPIController(p)


a, b = make(va), make(vb)
os.environ.pop("CCV_MPPI_KERNEL", None)
bad = 0
for it in range(n):
    s, xr, yr, yaw0 = inputs[it % len(inputs)]
    a.iterate_enqueue(s, p.dt, xr, yr, yaw0, 77, it)
    b.iterate_enqueue(s, p.dt, xr, yr, yaw0, 77, it)
    if it in (0, 1, n // 2, n - 1):
        ua, ub = a.get_nominal(), b.get_nominal()
        same = {"u*": np.array_equal(ua, ub), "costs": np.array_equal(a.read_costs(), b.read_costs()),
                "weights": np.array_equal(a.read_weights(), b.read_weights()),
                "controls": np.array_equal(a.read_controls(0, K), b.read_controls(0, K)),
                "states": np.array_equal(a.read_candidates(0, K // 16, 16), b.read_candidates(0, K // 16, 16))}
        print("%s K=%d iteration %d  %s vs %s: %s   finite %s" % (wl, K, it, va or "default", vb or "default", same, bool(np.all(np.isfinite(ua)))), flush=True)
        bad += sum(not v for v in same.values())
sys.exit(1 if bad else 0)
