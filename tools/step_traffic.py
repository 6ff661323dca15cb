#!/usr/bin/env python3
"""Fabric-side bytes per denoising step from rocprofv3 --pmc passes of `bench.py --no-split --no-cpu-baseline` at two step
counts: everything that is not a timed step (set-up, warm-up, the roofline legs) is the same in both runs and cancels in the
difference.  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 128-byte requests as 64, MI355X_MICROARCH.md).
Usage: step_traffic.py <root with {FETCH_SIZE,WRITE_SIZE}_{stepsA,stepsB}/ dirs> stepsA stepsB  -> JSON on stdout"""
import csv, glob, json, os, sys

root, sa, sb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])


def total(counter, steps):
    tot = 0.0
    for f in glob.glob(os.path.join(root, f"{counter}_{steps}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                tot += float(r["Counter_Value"])
    return tot


fetch = (total("FETCH_SIZE", sb) - total("FETCH_SIZE", sa)) / (sb - sa)
write = (total("WRITE_SIZE", sb) - total("WRITE_SIZE", sa)) / (sb - sa)
print(json.dumps({
    "what": "fabric-side traffic of ONE denoising step of bench.py's workload ([B=32,T=512], the dtype of the runs, single stream), from the "
            f"difference of two PMC runs with --steps {sa} and --steps {sb}",
    "command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python3 bench.py --no-split "
               "--no-cpu-baseline --steps <n>   (4 passes); tools/step_traffic.py",
    "FETCH_SIZE_KB_per_step": fetch, "WRITE_SIZE_KB_per_step": write,
    "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2; WRITE_SIZE exact; both in KiB",
    "bytes_per_step": int((2 * fetch + write) * 1024),
    "note": "counted at the L2 fabric side: Infinity-Cache hits are included, so this is an upper bound on HBM bytes",
}, indent=1))
