#!/usr/bin/env python3
"""Diagnostic (never shipped): per-phase cycle shares of k_rollout_coop from a -DCCV_STAMP build (_abl/lib_stamp.so).
Run on the GPU box:  python tools/stamps.py [K]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CCV_MPPI_LIB"] = os.path.join(ROOT, "_abl", "lib_stamp.so")
import numpy as np  # noqa: E402
import ccv_mppi_path_tracker_amd as amd  # noqa: E402
from ccv_mppi_path_tracker_amd import configs  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wl = sys.argv[2] if len(sys.argv) > 2 else "C2"
w = configs.workload(wl, num_samples=K)
p = w.params
px, py = amd.make_path(w.path)
s = np.zeros(p.nstate)
_, xr, yr, yaw = amd.calc_ref_path(px, py, 0.0, 0.0, p.v_ref, p.dt, p.resolution, p.horizon)
g = amd.MPPIController(p)
for it in range(5):
    g.iterate(s, p.dt, xr, yr, yaw[0], 1, it)
out = (C.c_ulonglong * 32)()
g.lib.ccv_mppi_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
g.lib.ccv_mppi_debug_stamps(g._h, out)
occ = (C.c_int * 4)()
g.lib.ccv_mppi_debug_occupancy(occ)
print("occupancy API: blocks/CU=%d numRegs=%d lds=%d scratch=%d" % tuple(occ))
names = ["P noise+u", "P recur", "P sincos", "P cost+xy", "P rest", "C dist", "barrier", "loop"]
a = np.array(list(out), dtype=np.float64).reshape(4, 8)
print("waves: 0 = dynamics + noise share, 1..3 = distance + noise share")
print("K=%d %s   cycles per wave (block 3), columns = waves 0..3" % (K, wl))
for i, n in enumerate(names):
    print("%-8s " % n + "  ".join("%8.0f" % v for v in a[:, i]))
print("total    " + "  ".join("%8.0f" % v for v in a.sum(axis=1)))

nb = min(4096, (K + 63) // 64)
blk = (C.c_ulonglong * (3 * nb))()
g.lib.ccv_mppi_debug_blocks.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
g.lib.ccv_mppi_debug_blocks(g._h, blk, nb)
b = np.array(list(blk), dtype=np.uint64).reshape(nb, 3)
t0 = b[:, 0].astype(np.float64); t1 = b[:, 1].astype(np.float64)
base = t0.min()
start_us = (t0 - base) / 100.0; end_us = (t1 - base) / 100.0     # s_memrealtime: 100 MHz
hw = (b[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64); xcc = (b[:, 2] >> np.uint64(32)).astype(np.int64)
cu = (hw >> 8) & 0xF; sh_ = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3
print("blocks %d: start min/median/max %.1f/%.1f/%.1f us, end min/median/max %.1f/%.1f/%.1f us, duration median %.1f us"
      % (nb, start_us.min(), np.median(start_us), start_us.max(), end_us.min(), np.median(end_us), end_us.max(),
         np.median(end_us - start_us)))
hist, edges = np.histogram(start_us, bins=12)
print("start-time histogram:", list(hist), " edges(us):", [round(e, 1) for e in edges])
key = xcc * 10000 + se * 1000 + sh_ * 100 + cu
uniq, counts = np.unique(key, return_counts=True)
print("distinct (xcc,se,sh,cu): %d; blocks per CU min/max: %d/%d; xcc histogram: %s" % (len(uniq), counts.min(), counts.max(),
      list(np.bincount(xcc, minlength=8))))
dur = end_us - start_us
for x in range(8):
    sel = xcc == x
    print("xcc %d: blocks %d  end median %.1f max %.1f  duration median %.1f" % (x, sel.sum(), np.median(end_us[sel]), end_us[sel].max(), np.median(dur[sel])))
order = np.argsort(end_us)
print("slowest 8 blocks (id, xcc, se, cu, end):", [(int(i), int(xcc[i]), int(se[i]), int(cu[i]), round(float(end_us[i]), 1)) for i in order[-8:]])
print("fastest 8 blocks:", [(int(i), int(xcc[i]), int(se[i]), int(cu[i]), round(float(end_us[i]), 1)) for i in order[:8]])
