#!/usr/bin/env python3
"""Diagnostic (never shipped): timeline of the four-wave kernel's workgroups from a -DCCV_DIAG build (_abl/lib_stamp.so, made by
`python tools/ablate.py stamp=-DCCV_DIAG`; the slots are listed in csrc/mppi_diag.h): when the pipeline is filled, when the loops
end, how long barrier and epilogue take.   python tools/stamps_r4.py [workload | dd1000 | sd1000]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CCV_MPPI_LIB", os.path.join(ROOT, "_abl", "lib_stamp.so"))
import numpy as np  # noqa: E402
import ccv_mppi_path_tracker_amd as amd  # noqa: E402
from ccv_mppi_path_tracker_amd import configs  # noqa: E402
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
if wl == "dd1000":
    w = configs.Workload("dd1000", configs.diff_drive_defaults(1000, 15), "sinusoid", "diff_drive K=1000 T=15 sinusoid")
elif wl == "sd1000":
    w = configs.Workload("sd1000", configs.steering_defaults(1000, 15), "sinusoid", "steering K=1000 T=15 sinusoid")
elif wl == "fb10000":
    w = configs.Workload("fb10000", configs.full_body_defaults(10000, 15), "dkan", "full body K=10000 T=15 dkan")
else:
    w = configs.workload(wl)
p = w.params
inputs = bench.script_inputs(amd, w, 64)
g = amd.MPPIController(p)
if os.environ.get("STAMP_RESIDENT"):   # the device-resident closed loop (pose and window from the frame in HBM)
    px, py = amd.make_path(w.path, p.resolution, length=400.0)
    s0 = np.zeros(p.nstate)
    s0[0], s0[1] = px[0], py[0]
    g.resident_set_path(px, py)
    g.resident_set_pose(s0)
    for it in range(1500):
        g.resident_step_enqueue(p.dt, 42, it, advance=it > 0)
else:
    for it in range(2000):
        s, xr, yr, yaw0 = inputs[it % len(inputs)]
        g.iterate_enqueue(s, p.dt, xr, yr, yaw0, 42, it)
g.synchronize()
NS = 16
nb = min(4096, (p.num_samples + 63) // 64)
blk = (C.c_ulonglong * (NS * nb))()
g.lib.ccv_mppi_debug_blocks.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
assert g.lib.ccv_mppi_debug_blocks(g._h, blk, nb) == 0
b = np.array(list(blk), dtype=np.float64).reshape(nb, NS)
t = (b - b[:, 0].min()) / 100.0     # us since the first workgroup's entry
names = ["entry", "first barrier passed", "staging barrier passed", "normals of block 0 published", "dynamics wave has them",
         "block 0 published (dynamics)", "dynamics loop end", "distance loop end", "wave 0 past the barrier",
         "wave 0 has its weight", "wave 0 done", "wave 2 has its weight", "wave 2 done", "wave 3 done",
         "distance wave through block 0", "noise loop end"]
if os.environ.get("STAMP_SAVE"):
    np.save(os.environ["STAMP_SAVE"], b)
print("%s: %d workgroups; us since the first workgroup's entry: mean  [min .. max]   | since its own entry: mean" % (wl, nb))
own = t - t[:, :1]
for i, n in enumerate(names):
    if b[:, i].max() == 0:
        continue
    print("  %2d %-32s %6.2f  [%6.2f .. %6.2f]   | %6.2f" % (i, n, t[:, i].mean(), t[:, i].min(), t[:, i].max(), own[:, i].mean()))

# who ends last?  by dispatch rank on the CU (workgroups go round-robin over the CUs: rank = block / #CUs) and by XCD (block % 8)
end = t[:, [10, 12, 13]].max(axis=1)
cus = 256
print("end of the workgroup (last of waves 0 / 2 / 3), us since the first entry: mean %.2f  max %.2f" % (end.mean(), end.max()))
for name, key in (("rank on its CU", np.arange(nb) // cus), ("XCD", np.arange(nb) % 8)):
    print("  by %s:" % name, "  ".join("%d: %.2f / %.2f / %.2f" % (v, t[key == v, 5].mean(), end[key == v].mean(), end[key == v].max()) for v in np.unique(key)),
          "  (block 0 published / end mean / end max)")
