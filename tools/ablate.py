#!/usr/bin/env python3
"""Experiment helper: build ablated variants of the library (never shipped) under _abl/ so that
bench.py can time them with CCV_MPPI_LIB=...  Usage: python tools/ablate.py NAME=-DFLAG1,-DFLAG2 ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ccv_mppi_path_tracker_amd import build  # noqa: E402

out_dir = os.path.join(ROOT, "_abl")
os.makedirs(out_dir, exist_ok=True)
for spec in sys.argv[1:]:
    name, _, flags = spec.partition("=")
    fl = [f for f in flags.split(",") if f]
    print(build.build(force=True, extra_flags=fl, out=os.path.join(out_dir, "lib_%s.so" % name)))
