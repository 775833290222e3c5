#!/usr/bin/env python3
"""Experiment helper: run bench.py once per _abl/lib_*.so (CCV_MPPI_LIB) and print kernel / iteration times.
Usage (on the GPU box): python tools/run_variants.py [bench args]"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for so in sorted(glob.glob(os.path.join(ROOT, "_abl", "lib_*.so"))):
    name = os.path.basename(so)[4:-3]
    env = dict(os.environ, CCV_MPPI_LIB=so)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + sys.argv[1:], env=env,
                       capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print("%-24s FAILED rc=%d %s" % (name, r.returncode, (r.stderr.strip().splitlines() or [""])[-1][:160]))
        continue
    d = json.loads(line[-1])
    rf = d["roofline"]
    print("%-24s ms_per_step %.4f  rollout_us %6.1f  iter_us %6.1f" % (name, d["ms_per_step"], rf["kernel_avg_us"],
                                                                      rf["iteration_avg_us"]), flush=True)
