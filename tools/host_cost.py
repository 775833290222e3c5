#!/usr/bin/env python3
"""Host-side cost of one ccv_mppi_iterate_enqueue call (argument set-up + launches), measured while the device queue is
kept short: N calls are timed from the host, then the queue is drained.  python tools/host_cost.py [workload]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ccv_mppi_path_tracker_amd as amd  # noqa: E402
from ccv_mppi_path_tracker_amd import configs  # noqa: E402

w = configs.workload(sys.argv[1] if len(sys.argv) > 1 else "C2")
p = w.params
px, py = amd.make_path(w.path)
s = np.zeros(p.nstate)
_, xr, yr, yaw = amd.calc_ref_path(px, py, 0.0, 0.0, p.v_ref, p.dt, p.resolution, p.horizon)
g = amd.MPPIController(p)
st = torch.cuda.Stream()
g.set_stream(st.cuda_stream)
for it in range(20):
    g.iterate_enqueue(s, p.dt, xr, yr, yaw[0], 1, it)
g.synchronize()
for n in (8, 64, 512, 2048, 512, 2048, 512, 200, 200, 200):
    t0 = time.perf_counter()
    for it in range(n):
        g.iterate_enqueue(s, p.dt, xr, yr, yaw[0], 1, 100 + it)
    t1 = time.perf_counter()
    g.synchronize()
    t2 = time.perf_counter()
    print("n=%4d: host %.1f us/call while enqueueing, %.1f us/iteration end to end" % (n, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n))
print("cpus:", os.cpu_count(), "loadavg:", os.getloadavg())
