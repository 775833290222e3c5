#!/usr/bin/env python3
"""Blocking ccv_mppi_iterate latency (host pointers in, u* out through pinned memory) vs enqueue-only throughput."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ccv_mppi_path_tracker_amd as amd
from ccv_mppi_path_tracker_amd import configs
for wl in ("C2", "C3", "C4"):
    w = configs.workload(wl); p = w.params
    px, py = amd.make_path(w.path)
    s = np.zeros(p.nstate); s[:2] = px[0], py[0]
    _, xr, yr, yaw = amd.calc_ref_path(px, py, s[0], s[1], p.v_ref, p.dt, p.resolution, p.horizon)
    g = amd.MPPIController(p)
    for i in range(20): g.iterate(s, p.dt, xr, yr, yaw[0], 1, i, want_stats=False)
    t0 = time.perf_counter(); n = 200
    for i in range(n): g.iterate(s, p.dt, xr, yr, yaw[0], 1, 20 + i, want_stats=False)
    tb = (time.perf_counter() - t0) / n
    g.synchronize(); t0 = time.perf_counter()
    for i in range(n): g.iterate_enqueue(s, p.dt, xr, yr, yaw[0], 1, 300 + i)
    g.synchronize(); te = (time.perf_counter() - t0) / n
    print("%s blocking iterate %.1f us   enqueue-only %.1f us per iteration" % (wl, tb * 1e6, te * 1e6))
