#!/usr/bin/env python3
"""Groups a rocprofv3 --kernel-trace CSV by (kernel, grid size): per-shape launch count, average and total time.
Usage: summarize_trace.py <kernel_trace.csv> [min_start_fraction]"""
import collections, csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "")
    grid = r.get("Grid_Size") or "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
    wg = r.get("Workgroup_Size") or r.get("Workgroup_Size_X", "?")
    key = (name, grid, wg)
    agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
print(f"{'kernel':60s} {'grid':>14s} {'wg':>5s} {'calls':>6s} {'avg_us':>9s} {'total_ms':>9s} {'%':>6s}")
for (name, grid, wg), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{name[:60]:60s} {grid:>14s} {wg:>5s} {len(v):6d} {sum(v)/len(v)/1e3:9.1f} {sum(v)/1e6:9.3f} {100*sum(v)/tot:6.2f}")
