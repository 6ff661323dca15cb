#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection CSVs per launch for the kernels whose name contains a substring.
Usage: pmc_summary.py <kernel-substring>[@<grid_x>x<grid_y>] <dir-with-*counter_collection.csv> [...more dirs]   -> JSON on stdout
(the optional @grid keeps only launches of that grid size in work-items, e.g. "conv_gemm_big_kernel<dn::BF16, 4>@32768x3")"""
import collections, csv, glob, json, os, sys

sub, _, grid = sys.argv[1].partition("@")
gx, _, gy = grid.partition("x")
acc = collections.defaultdict(list)
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if sub not in r["Kernel_Name"]:
                continue
            if grid and "Grid_Size_X" in r and (r["Grid_Size_X"] != gx or (gy and r.get("Grid_Size_Y", "") != gy)):
                continue
            if grid and "Grid_Size_X" not in r and int(r["Grid_Size"]) != int(gx) * int(gy or 1):  # one column: the product
                continue
            per_dispatch[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (_, name), v in per_dispatch.items():
            acc[name].append(v)
print(json.dumps({k: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for k, v in sorted(acc.items())}, indent=1))
