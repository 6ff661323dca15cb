#!/usr/bin/env python3
"""Prints, for every launch of a kernel whose name contains <substring>, the launches before and after it (a rocprofv3 --kernel-trace
CSV, in start order): where in the step does it sit?  Usage: trace_neighbors.py <kernel_trace.csv> <substring> [max matches]"""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "")[:70]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 60
skip = len(rows) // 2  # steady state
n = 0
for i in range(skip, len(rows)):
    if sys.argv[2] in rows[i]["Kernel_Name"]:
        d = (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
        print(f"{i:6d} {name(rows[i-1]):60s} -> [{name(rows[i])} {rows[i]['Grid_Size_X']} {d:.1f}us] -> {name(rows[i+1]) if i + 1 < len(rows) else ''}")
        n += 1
        if n >= lim:
            break
