#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter CSVs per kernel: python tools/pmc_sum.py DIR [kernel-substring]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if pat and pat not in k:
            continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[k].add(row["Dispatch_Id"])
for k in acc:
    print(k, "dispatches", len(calls[k]))
    for c, v in sorted(acc[k].items()):
        print(f"   {c:28s} {v:.6g}")
