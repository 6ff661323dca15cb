#!/usr/bin/env python3
"""Short hash of the kernel sources (diffnorm_amd/csrc/*.hip, *.h + the C header): profiles record it when they are collected and
bench.py compares it with the build it runs, so a committed counter pass that no longer belongs to the committed kernels is
visible in the JSON line (`roofline.traffic_profile_matches_build`)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_id() -> str:
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(ROOT, "diffnorm_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "diffnorm_amd", "csrc", "*.h")) +
                   [os.path.join(ROOT, "include", "diffnorm_hip.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


if __name__ == "__main__":
    print(build_id())
