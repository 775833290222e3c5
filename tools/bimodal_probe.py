#!/usr/bin/env python3
"""Diagnostic: does the rollout-kernel time of one workload change between handles of ONE process (i.e. with the buffers'
placement) or only between processes?   python tools/bimodal_probe.py [workload] [handles]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ccv_mppi_path_tracker_amd as amd  # noqa: E402
from ccv_mppi_path_tracker_amd import configs  # noqa: E402
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
n_handles = int(sys.argv[2]) if len(sys.argv) > 2 else 6
w = configs.workload(wl)
p = w.params
inputs = bench.script_inputs(amd, w, 64)
keep = []
for h in range(n_handles):
    g = amd.MPPIController(p)
    for it in range(600):
        s, xr, yr, yaw0 = inputs[it % len(inputs)]
        g.iterate_enqueue(s, p.dt, xr, yr, yaw0, 42, it)
    g.synchronize()
    res = []
    for rep in range(3):
        g.timing_enable(True, every=1)
        g.timing_read(reset=True)
        for it in range(256):
            s, xr, yr, yaw0 = inputs[it % len(inputs)]
            g.iterate_enqueue(s, p.dt, xr, yr, yaw0, 42, 1000 + it)
        g.synchronize()
        r, i, n = g.timing_read(reset=True)
        g.timing_enable(False)
        res.append(r / n)
    print("handle %d: kernel us %s" % (h, " ".join("%.2f" % x for x in res)), flush=True)
    if os.environ.get("PROBE_KEEP_ALL") or h % 2 == 0:
        keep.append(g)      # (keeps its buffers: the next handle gets other memory)
    else:
        g.close()
