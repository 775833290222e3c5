#!/usr/bin/env python3
"""Diagnostic (never shipped): per-phase cycle shares and workgroup timelines of the rollout kernels from a -DCCV_STAMP build (_abl/lib_stamp.so).
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
NB2B = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # back-to-back launches before the read-out (steady-state clock)
for it in range(NB2B):
    g.iterate_enqueue(s, p.dt, xr, yr, yaw[0], 1, 5 + it)
g.synchronize()
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
blk = (C.c_ulonglong * (6 * nb))()
g.lib.ccv_mppi_debug_blocks.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
g.lib.ccv_mppi_debug_blocks(g._h, blk, nb)
b = np.array(list(blk), dtype=np.uint64).reshape(nb, 6)
t0 = b[:, 0].astype(np.float64)
base = t0.min()
start_us = (t0 - base) / 100.0                                   # s_memrealtime: 100 MHz
end0_us = (b[:, 1].astype(np.float64) - base) / 100.0            # loop end, wave 0 / wave 1; kernel end (after the epilogue)
end1_us = (b[:, 3].astype(np.float64) - base) / 100.0
fin_us = (b[:, 5].astype(np.float64) - base) / 100.0
end_us = np.maximum(end0_us, end1_us)


def decode(word):
    hw = (word & np.uint64(0xFFFFFFFF)).astype(np.int64)
    return dict(xcc=(word >> np.uint64(32)).astype(np.int64), cu=(hw >> 8) & 0xF, sh=(hw >> 12) & 0x1, se=(hw >> 13) & 0x7,
                simd=(hw >> 4) & 0x3, wave=hw & 0xF)


h0, h1 = decode(b[:, 2]), decode(b[:, 4])
xcc = h0["xcc"]
dur = end_us - start_us
print("blocks %d: start max %.1f us; loop end min/median/max %.1f/%.1f/%.1f us; kernel end median/max %.1f/%.1f us; "
      "epilogue median %.1f us" % (nb, start_us.max(), end_us.min(), np.median(end_us), end_us.max(), np.median(fin_us),
                                   fin_us.max(), np.median(fin_us - end0_us)))
d3 = float(end0_us[3] - start_us[3])
print("block 3: %.1f us by s_memrealtime, %.0f s_memtime ticks in the loop => shader clock >= %.2f GHz"
      % (d3, a[0].sum(), a[0].sum() / d3 / 1e3))
print("loop duration histogram:", np.histogram(dur, bins=10))
# SIMD placement: how many waves of the kernel share a SIMD with this block's waves?
cuid = xcc * 1000 + h0["se"] * 100 + h0["sh"] * 50 + h0["cu"]
simd_key0 = cuid * 4 + h0["simd"]
simd_key1 = cuid * 4 + h1["simd"]
allk = np.concatenate([simd_key0, simd_key1])
uk, cnt = np.unique(allk, return_counts=True)
print("waves per SIMD histogram (over %d SIMDs used):" % len(uk), np.bincount(cnt))
load = dict(zip(uk.tolist(), cnt.tolist()))
l0 = np.array([load[k] for k in simd_key0.tolist()]); l1 = np.array([load[k] for k in simd_key1.tolist()])
same = (h0["simd"] == h1["simd"])
print("blocks whose two waves share a SIMD: %d" % same.sum())
for lo in sorted(set((np.maximum(l0, l1)).tolist())):
    sel = np.maximum(l0, l1) == lo
    print("max waves on a SIMD used by the block = %d: %d blocks, loop duration median %.1f us" % (lo, sel.sum(), np.median(dur[sel])))
print("same-SIMD blocks: median %.1f; split blocks: median %.1f" % (np.median(dur[same]) if same.any() else -1,
                                                                   np.median(dur[~same]) if (~same).any() else -1))
ep = fin_us - end0_us
print("epilogue (kernel end - loop end, wave 0) histogram:", np.histogram(ep, bins=10))
order = np.argsort(fin_us)
print("last 6 blocks to end: ", [(int(i), round(float(end0_us[i]), 1), round(float(fin_us[i]), 1)) for i in order[-6:]])

# placement of the workgroups of a few CUs: (block id, SIMD / wave slot of wave 0, SIMD / wave slot of wave 1, loop duration)
print("placement (first 6 CUs): block: w0 simd/slot, w1 simd/slot, loop us")
seen = {}
for i in range(nb):
    seen.setdefault(int(cuid[i]), []).append(i)
for cu_k in sorted(seen)[:6]:
    print("  cu %d: " % cu_k + "; ".join("%d: %d/%d %d/%d %.1f" % (i, h0["simd"][i], h0["wave"][i], h1["simd"][i], h1["wave"][i], dur[i])
                                        for i in seen[cu_k]))
if os.environ.get("CCV_MPPI_KERNEL", "") != "pc" and wl == "C2":
    raw = np.array(list(out), dtype=np.float64)
    for name, o in (("block 3 (first on its CU)", 0), ("block 771 (last on its CU)", 8)):
        print("three-wave kernel, %s: work / loop cycles  producer %.0f/%.0f  distance %.0f/%.0f  store %.0f/%.0f"
              % ((name,) + tuple(raw[o:o + 6])))
