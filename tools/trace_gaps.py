#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV (single-stream runs): per-step busy / gap totals.
Usage: trace_gaps.py <kernel_trace.csv> [skip_fraction]"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50]) for r in csv.DictReader(open(sys.argv[1]))))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]  # steady state: graph replays
busy = sum(e - s for s, e, _ in rows)
gaps = [max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1)]
span = rows[-1][1] - rows[0][0]
import collections
hist = collections.Counter(min(g // 1000, 20) for g in gaps)
print(f"kernels {len(rows)}  span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms ({100*busy/span:.1f} %)  gaps {sum(gaps)/1e6:.2f} ms, mean {sum(gaps)/len(gaps)/1e3:.2f} us")
print("gap histogram (us: count):", dict(sorted(hist.items())))
big = sorted(((g, rows[i][2], rows[i + 1][2]) for i, g in enumerate(gaps)), reverse=True)[:8]
for g, a, b in big: print(f"  {g/1e3:8.1f} us after {a} before {b}")
