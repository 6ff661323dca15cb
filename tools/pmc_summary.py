#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection CSVs per launch for the kernels whose name contains a substring.
Usage: pmc_summary.py <kernel-substring> <dir-with-*counter_collection.csv> [...more dirs]   -> JSON on stdout"""
import collections, csv, glob, json, os, sys

sub = sys.argv[1]
acc = collections.defaultdict(list)
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if sub not in r["Kernel_Name"]:
                continue
            per_dispatch[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (_, name), v in per_dispatch.items():
            acc[name].append(v)
print(json.dumps({k: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for k, v in sorted(acc.items())}, indent=1))
