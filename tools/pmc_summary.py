#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSVs (one row per dispatch x counter) per kernel: mean counter value per launch."""
import collections, csv, glob, os, sys
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "hbmpc" not in k and "mfma" not in k:
            continue
        agg[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:26s} mean/launch {sum(v)/len(v):16.1f}   launches {len(v)}")
