#!/usr/bin/env python3
"""Per-(kernel, grid) fabric traffic AND rate from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command: bytes per
launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB (gfx950 read correction), rate = bytes / the launch's duration in the same pass.
A kernel that moves > ~3.5 TB/s through the fabric is bandwidth-bound: fewer bytes is the way to make it faster.
Usage: traffic_rate_by_kernel.py <dir with FETCH_SIZE/ and WRITE_SIZE/ sub-directories> [min total ms]"""
import collections, csv, glob, os, re, sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0, "ns": 0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(root, c, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            key = (re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", ""), r["Grid_Size"])
            acc[key][c] += float(r["Counter_Value"])
            if c == "FETCH_SIZE" and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                acc[key]["n"] += 1
                acc[key]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
rows = []
for k, v in acc.items():
    if not v["n"]:
        continue
    b = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    rows.append((v["ns"], k, v, b))
tot_ns = sum(r[0] for r in rows)
print(f"{'kernel':56s} {'grid':>9s} {'calls':>6s} {'avg us':>8s} {'% time':>7s} {'MB/launch':>10s} {'TB/s':>6s}")
for ns, k, v, b in sorted(rows, reverse=True)[:40]:
    print(f"{k[0][:56]:56s} {k[1]:>9s} {v['n']:6d} {ns / v['n'] / 1e3:8.1f} {100 * ns / tot_ns:7.2f} {b / v['n'] / 1e6:10.1f} {b / max(ns, 1) / 1e3:6.2f}")
