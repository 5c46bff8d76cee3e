#!/usr/bin/env python3
"""Shrink rocprofv3 counter_collection.csv files in place to one row per (kernel, counter) holding the
mean over the dispatches - all tools/make_profile_summary.py needs, and a few KB instead of tens of MB.
    python3 tools/pmc_reduce.py DIR"""
import csv
import glob
import sys
from collections import defaultdict

for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Dispatches"])
        for (k, c), v in sorted(acc.items()):
            w.writerow([k, c, sum(v) / len(v), len(v)])
