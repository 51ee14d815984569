#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel name, the mean counter value per dispatch."""
import csv, glob, sys, collections, os
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:60]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"    {c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
