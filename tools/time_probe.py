#!/usr/bin/env python3
"""Diagnostic: ONE handle, ONE stream, the same 256 iterations timed over and over -- does the rollout-kernel time change with time?
python tools/time_probe.py [workload] [repeats] [pause_ms]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ccv_mppi_path_tracker_amd as amd  # noqa: E402
from ccv_mppi_path_tracker_amd import configs  # noqa: E402
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
pause = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
w = configs.workload(wl)
p = w.params
inputs = bench.script_inputs(amd, w, 64)
g = amd.MPPIController(p)
out = []
for rep in range(reps):
    g.timing_enable(True, every=1)
    g.timing_read(reset=True)
    for it in range(256):
        s, xr, yr, yaw0 = inputs[it % len(inputs)]
        g.iterate_enqueue(s, p.dt, xr, yr, yaw0, 42, 1000 + it)
    g.synchronize()
    r, i, cnt = g.timing_read(reset=True)
    g.timing_enable(False)
    out.append(r / cnt)
    if pause:
        time.sleep(pause * 1e-3)
print(wl, "pause %g ms:" % pause, " ".join("%.1f" % x for x in out))
