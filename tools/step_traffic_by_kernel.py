#!/usr/bin/env python3
"""Per-kernel fabric traffic of one denoising step from the PMC passes of tools/collect_step_traffic_r02.sh: the difference of the
--steps 30 and --steps 10 runs, grouped by (kernel, grid).  bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB (gfx950 read correction).
Usage: step_traffic_by_kernel.py <root with {FETCH_SIZE,WRITE_SIZE}_{10,30}/ dirs>"""
import collections, csv, glob, os, re, sys

root = sys.argv[1]


def table(counter, steps):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(root, f"{counter}_{steps}", "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "")
            key = (name, r["Grid_Size"])
            acc[key][0] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                acc[key][1] += 1
    return acc


rows = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    a, b = table(c, 10), table(c, 30)
    for k in b:
        kb = (b[k][0] - a.get(k, [0, 0])[0]) / 20.0
        n = (b[k][1] - a.get(k, [0, 0])[1]) / 20.0
        rows.setdefault(k, {})[c] = kb
        rows[k]["launches"] = n
tot = sum((2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) for v in rows.values())
print(f"{'kernel':58s} {'grid':>9s} {'launches/step':>13s} {'read MB':>9s} {'write MB':>9s} {'MB/launch':>10s} {'% of step':>9s}")
for k, v in sorted(rows.items(), key=lambda kv: -(2 * kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0))):
    rd, wr = 2 * v.get("FETCH_SIZE", 0) * 1024 / 1e6, v.get("WRITE_SIZE", 0) * 1024 / 1e6
    if rd + wr < 1.0:
        continue
    n = max(v["launches"], 1e-9)
    print(f"{k[0][:58]:58s} {k[1]:>9s} {v['launches']:13.1f} {rd:9.1f} {wr:9.1f} {(rd + wr) / n:10.1f} {100 * (rd + wr) * 1e6 / 1024 / tot:9.2f}")
print(f"total {tot * 1024 / 1e9:.2f} GB per step")
