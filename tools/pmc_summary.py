#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel, mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(path)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "msmv" not in k:
                continue
            print(k[:60])
            for c, v in sorted(cs.items()):
                print(f"    {c:28s} {sum(v) / len(v):16.0f}  (n={len(v)})")
