#!/usr/bin/env python3
"""Diagnostic (never shipped): timeline of the four-wave kernel's workgroups from a -DCCV_STAMP build (_abl/lib_stamp.so, made by
`python tools/ablate.py stamp=-DCCV_STAMP`): when the pipeline is filled, when the loops end, how long barrier and epilogue take;
with a -DCCV_STAMP=2 build and STAMP_SET=fill: the start of the kernel in detail.   python tools/stamps_r4.py [workload]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CCV_MPPI_LIB"] = os.path.join(ROOT, "_abl", "lib_stamp.so")
import numpy as np  # noqa: E402
import ccv_mppi_path_tracker_amd as amd  # noqa: E402
from ccv_mppi_path_tracker_amd import configs  # noqa: E402
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
w = configs.workload(wl)
p = w.params
inputs = bench.script_inputs(amd, w, 64)
g = amd.MPPIController(p)
for it in range(2000):
    s, xr, yr, yaw0 = inputs[it % len(inputs)]
    g.iterate_enqueue(s, p.dt, xr, yr, yaw0, 42, it)
g.synchronize()
nb = min(4096, (p.num_samples + 63) // 64)
blk = (C.c_ulonglong * (6 * nb))()
g.lib.ccv_mppi_debug_blocks.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
g.lib.ccv_mppi_debug_blocks(g._h, blk, nb)
b = np.array(list(blk), dtype=np.float64).reshape(nb, 6)
t = (b - b[:, 0].min()) / 100.0     # us since the first workgroup's entry
names = ["entry", "block 0 published (dynamics)", "dynamics loop end", "distance loop end", "past the barrier", "epilogue end"]
if os.environ.get("STAMP_SET") == "epi":   # a -DCCV_STAMP=4 build
    names = ["wave 0 past the barrier", "wave 0 has its weight", "wave 0 done", "wave 2 has its weight", "wave 2 done", "wave 3 done"]
if os.environ.get("STAMP_SET") == "fill":   # a -DCCV_STAMP=2 build
    names = ["entry", "first barrier passed", "staging barrier passed", "block 0's normals published", "dynamics wave has them",
             "block 0 published (dynamics)"]
print("%s: %d workgroups; us since the first workgroup's entry: mean  [min .. max]" % (wl, nb))
for i, n in enumerate(names):
    print("  %-30s %6.2f  [%6.2f .. %6.2f]" % (n, t[:, i].mean(), t[:, i].min(), t[:, i].max()))
d = t - t[:, :1]
print("per workgroup, us since its own entry: mean")
for i, n in enumerate(names[1:], 1):
    print("  %-30s %6.2f" % (n, d[:, i].mean()))
if not os.environ.get("STAMP_SET"):
    print("  barrier wait after the distance loop %5.2f   epilogue %5.2f" % ((t[:, 4] - t[:, 3]).mean(), (t[:, 5] - t[:, 4]).mean()))
