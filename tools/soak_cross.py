#!/usr/bin/env python3
"""Diagnostic (run by hand on a GPU box): how far do the default kernel and the two-wave kernel drift apart over n iterations
of the scripted loop?  (Same samples; costs differ in the order of a sample's terms, i.e. by ~1e-16 per iteration; the warm
start feeds back.)   python tools/soak_cross.py WORKLOAD K n1 n2 ..."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ccv_mppi_path_tracker_amd as amd  # noqa: E402
from ccv_mppi_path_tracker_amd import configs  # noqa: E402
import bench  # noqa: E402

wl, K = sys.argv[1], int(sys.argv[2])
w = configs.workload(wl, num_samples=K)
p = w.params
inputs = bench.script_inputs(amd, w, 64)
os.environ.pop("CCV_MPPI_KERNEL", None)
a = amd.MPPIController(p)
os.environ["CCV_MPPI_KERNEL"] = "pc"
b = amd.MPPIController(p)
done = 0
for n in [int(x) for x in sys.argv[3:]]:
    for it in range(done, n):
        s, xr, yr, yaw0 = inputs[it % len(inputs)]
        a.iterate_enqueue(s, p.dt, xr, yr, yaw0, 77, it)
        b.iterate_enqueue(s, p.dt, xr, yr, yaw0, 77, it)
    done = n
    ua, ub = a.get_nominal(), b.get_nominal()
    ca, cb = a.read_costs(), b.read_costs()
    print("%s K=%d after %6d iterations: u* rel diff %.2e   costs rel diff %.2e   controls equal %s" % (
        wl, K, n, np.max(np.abs(ua - ub)) / np.max(np.abs(ub)), np.max(np.abs(ca - cb) / np.abs(cb)),
        np.array_equal(a.read_controls(0, 64), b.read_controls(0, 64))))
