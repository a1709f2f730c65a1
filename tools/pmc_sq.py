#!/usr/bin/env python3
"""Summarise an SQ counter pass (rocprofv3 --pmc SQ_... -- python3 tools/kbench.py ...): per-kernel averages."""
import csv, glob, os, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "nsg" not in k:
            continue
        acc[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    n = len(next(iter(cs.values())))
    print(k, f"({n} dispatches)")
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.1f}")
