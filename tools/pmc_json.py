#!/usr/bin/env python3
"""profiles/<tag>_<workload>_pmc.json from the rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of tools/profile_r03.sh: HBM
bytes per launch of the rollout kernel, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950
(WRITE_SIZE exact; FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streams: doubled).  bench.py reads the
latest of these for `roofline.traffic`.  usage: python tools/pmc_json.py <gpurun_out/tag dir> <tag>"""
import csv
import glob
import json
import os
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
ALG = {"C2": (8 * (2 * 49 * 2 + 2 * 50 + 2)) * 65536, "C3": (8 * (2 * 49 * 3 + 2 * 50 + 2)) * 65536, "C4": (8 * (2 * 79 * 5 + 2 * 80 + 2)) * 131072}


def mean_counter(d, counter):
    vals, name = [], None
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "rollout" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
                name = r["Kernel_Name"].split("(")[0]
    return (sum(vals) / len(vals) if vals else None), name, len(vals)


for w in ("C2", "C3", "C4"):
    wr, name, n = mean_counter(os.path.join(out_dir, "pmc_%s_WRITE_SIZE" % w), "WRITE_SIZE")
    fe, _, _ = mean_counter(os.path.join(out_dir, "pmc_%s_FETCH_SIZE" % w), "FETCH_SIZE")
    if wr is None or fe is None:
        continue
    d = {"workload": w, "kernel": name, "launches_averaged": n, "WRITE_SIZE_KB": round(wr), "FETCH_SIZE_KB_raw": round(fe),
         "correction": "gfx950: WRITE_SIZE exact, FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md "
                       "'FETCH_SIZE reports exactly 1/2'; calibrated for this code's 8-byte-per-lane pattern with tools/microbench/hbm_calib.hip): doubled",
         "hbm_bytes_per_launch": int((wr + 2 * fe) * 1024), "algorithmic_bytes_per_launch": ALG[w],
         "command": "tools/profile_r03.sh %s: rocprofv3 --pmc WRITE_SIZE (and, in its own run, --pmc FETCH_SIZE) --output-format csv -- python3 bench.py "
                    "--workload %s --steps 40 --warmup 5 --no-cpu-baseline --no-other-workloads --no-defaults-leg --no-kernel-events --no-closed-loop-leg" % (tag, w)}
    json.dump(d, open(os.path.join(out_dir, "%s_pmc.json" % w), "w"), indent=1)
    print(w, d["hbm_bytes_per_launch"], "vs algorithmic", ALG[w], "ratio %.3f" % (d["hbm_bytes_per_launch"] / ALG[w]))
