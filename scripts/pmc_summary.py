#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel (name prefix), per counter: mean per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "k_fwd"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if want not in name:
            continue
        acc[name[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
