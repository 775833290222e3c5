#!/usr/bin/env python3
"""Soak / race screen (run by hand on a GPU box, not collected by pytest): the same scripted closed-loop-like sequence of
iterations twice on fresh handles; warm start, costs and candidate states must come out bit-identical.  Any race in the
LDS hand-offs, the store wave or the deferred division would show as a difference after thousands of launches.
  python tests/soak_determinism.py [iterations]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ccv_mppi_path_tracker_amd as amd  # noqa: E402
from ccv_mppi_path_tracker_amd import configs  # noqa: E402
import bench  # noqa: E402

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
ok = True
for wl, K in (("C2", 65536), ("C3", 65536), ("C4", 32768), ("C2", 1000), ("C4", 10000), ("C3", 1000)):
    w = configs.workload(wl, num_samples=K)
    p = w.params
    inputs = bench.script_inputs(amd, w, 64)
    res = []
    for rep in range(2):
        g = amd.MPPIController(p)
        for it in range(n_it):
            s, xr, yr, yaw0 = inputs[it % len(inputs)]
            g.iterate_enqueue(s, p.dt, xr, yr, yaw0, 77, it)
        g.synchronize()
        res.append((g.get_nominal().copy(), g.read_costs().copy(), g.read_candidates(0, 64, max(1, K // 64)).copy()))
        g.close()
    same = all(np.array_equal(a, b) for a, b in zip(res[0], res[1]))
    finite = bool(np.all(np.isfinite(res[0][0])))
    # ... and against the two-wave kernel (a barrier per time block, no sequence-number hand-offs) over the first 16
    # iterations: the same samples, costs equal up to the order of a sample's terms.  (Not over the whole run: the warm start
    # feeds back and the MPPI map amplifies rounding differences by ~3 % per iteration -- 1e-15 after one iteration, 1e-13
    # after 100, order one after a few thousand, identically for the barrier and the sequence-number version of the
    # three-wave kernel: tools/soak_cross.py.)
    pair = []
    for kern in (None, "pc"):
        if kern:
            os.environ["CCV_MPPI_KERNEL"] = kern
        g = amd.MPPIController(p)
        for it in range(16):
            s, xr, yr, yaw0 = inputs[it % len(inputs)]
            g.iterate_enqueue(s, p.dt, xr, yr, yaw0, 77, it)
        pair.append((g.get_nominal().copy(), g.read_candidates(0, 64, max(1, K // 64)).copy()))
        g.close()
        os.environ.pop("CCV_MPPI_KERNEL", None)
    du = float(np.max(np.abs(pair[0][0] - pair[1][0])) / max(np.max(np.abs(pair[1][0])), 1e-300))
    dxy = float(np.max(np.abs(pair[0][1] - pair[1][1])))
    # (full body with the C4 weights has an effective sample size of order one: there the map amplifies a rounding difference
    #  by ~2.5 per iteration -- 2e-13 after one iteration, 2e-7 after 16, order one after 24: tools/soak_cross.py C4 10000 ...)
    tol = 1e-5 if p.model == "full_body" else 1e-9
    cross = du < tol and dxy < tol
    print("%s K=%d: %d iterations twice: bit-identical %s, finite %s; 16 iterations vs the two-wave kernel: u* rel %.1e, states abs %.1e" % (
        wl, K, n_it, same, finite, du, dxy))
    ok = ok and same and finite and cross
sys.exit(0 if ok else 1)
