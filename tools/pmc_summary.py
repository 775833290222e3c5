#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean per launch for kernels whose name contains a pattern.
usage: python tools/pmc_summary.py <dir-or-csv> [pattern]"""
import collections
import csv
import glob
import os
import sys

src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "rollout"
files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][-48:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k)
    for c in sorted(v):
        x = v[c]
        print("   %-28s %14.0f  (n=%d)" % (c, sum(x) / len(x), len(x)))
