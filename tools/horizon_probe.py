#!/usr/bin/env python3
"""Diagnostic: rollout-kernel time against the horizon -- what does the last, partial time block of a horizon cost?
python tools/horizon_probe.py [workload] H1 H2 ..."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dataclasses  # noqa: E402

import ccv_mppi_path_tracker_amd as amd  # noqa: E402
from ccv_mppi_path_tracker_amd import configs  # noqa: E402
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
for H in [int(x) for x in sys.argv[2:]] or [48, 49, 50, 51, 57, 58]:
    w0 = configs.workload(wl)
    w = dataclasses.replace(w0, params=w0.params.with_(horizon=H))
    p = w.params
    inputs = bench.script_inputs(amd, w, 64)
    g = amd.MPPIController(p)
    for it in range(3000):
        s, xr, yr, yaw0 = inputs[it % len(inputs)]
        g.iterate_enqueue(s, p.dt, xr, yr, yaw0, 42, it)
    g.synchronize()
    res = []
    for rep in range(3):
        g.timing_enable(True, every=1)
        g.timing_read(reset=True)
        for it in range(256):
            s, xr, yr, yaw0 = inputs[it % len(inputs)]
            g.iterate_enqueue(s, p.dt, xr, yr, yaw0, 42, 5000 + it)
        g.synchronize()
        r, i, n = g.timing_read(reset=True)
        g.timing_enable(False)
        res.append(r / n)
    print("%s H = %d (%d steps = %d full blocks + %d): kernel us %s" % (wl, H, H - 1, (H - 1) // 8, (H - 1) % 8,
                                                                       " ".join("%.2f" % x for x in res)), flush=True)
    g.close()
