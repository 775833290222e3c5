#!/usr/bin/env python3
"""Diagnostic: ONE handle, several streams -- does the rollout-kernel time depend on the stream (hardware queue) it is launched on?
python tools/stream_probe.py [workload] [streams]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ccv_mppi_path_tracker_amd as amd  # noqa: E402
from ccv_mppi_path_tracker_amd import configs  # noqa: E402
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
w = configs.workload(wl)
p = w.params
inputs = bench.script_inputs(amd, w, 64)
g = amd.MPPIController(p)
streams = [None] + [torch.cuda.Stream() for _ in range(n)]
for si, st in enumerate(streams):
    g.set_stream(0 if st is None else st.cuda_stream)
    for it in range(400):
        s, xr, yr, yaw0 = inputs[it % len(inputs)]
        g.iterate_enqueue(s, p.dt, xr, yr, yaw0, 42, it)
    g.synchronize()
    g.timing_enable(True, every=1)
    g.timing_read(reset=True)
    for it in range(256):
        s, xr, yr, yaw0 = inputs[it % len(inputs)]
        g.iterate_enqueue(s, p.dt, xr, yr, yaw0, 42, 1000 + it)
    g.synchronize()
    r, i, cnt = g.timing_read(reset=True)
    g.timing_enable(False)
    print("stream %2d (%s): kernel %.2f us" % (si, "handle's own" if st is None else hex(st.cuda_stream), r / cnt), flush=True)
