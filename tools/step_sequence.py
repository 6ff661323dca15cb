#!/usr/bin/env python3
"""Per-position timeline of one denoising step from a rocprofv3 --kernel-trace CSV: the launches between two consecutive
`ddim_step_kernel`s of the steady chain, averaged over the steps of the trace (duration and the idle gap before each launch).
Usage: step_sequence.py <kernel_trace.csv> [marker kernel substring]"""
import collections, csv, re, sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "ddim_step_kernel"
name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "")
cuts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
segs = [rows[a + 1: b + 1] for a, b in zip(cuts, cuts[1:])]
sig = lambda s: tuple((name(r), r["Grid_Size_X"], r["Grid_Size_Y"]) for r in s)
modal, _ = collections.Counter(sig(s) for s in segs).most_common(1)[0]
segs = [s for s in segs if sig(s) == modal]
n = len(modal)
dur, gap = [0.0] * n, [0.0] * n
for s in segs:
    for i, r in enumerate(s):
        dur[i] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if i:
            gap[i] += max(0, int(r["Start_Timestamp"]) - int(s[i - 1]["End_Timestamp"]))
print(f"{len(segs)} steps of {n} launches; per step: kernels {sum(dur)/len(segs)/1e3:.1f} us, gaps {sum(gap)/len(segs)/1e3:.1f} us")
for i, (nm, gx, gy) in enumerate(modal):
    print(f"{i:3d} {nm[:58]:58s} {gx:>8s}x{gy:<3s} {dur[i]/len(segs)/1e3:8.1f} us  gap {gap[i]/len(segs)/1e3:6.1f}")
