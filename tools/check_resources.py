#!/usr/bin/env python3
"""Compiles the HIP sources for gfx950 with -save-temps and fails if any kernel spills VGPRs or uses scratch
(a spill in a hand-scheduled K loop also breaks its counted vmcnt waits: scratch loads share that counter).
Usage: check_resources.py [file.hip ...]   (default: gemm.hip attention.hip)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "diffnorm_amd", "csrc")
FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def kernels(asm_text):
    for blk in asm_text.split("  - .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
        yield name, get("vgpr_count"), get("vgpr_spill_count"), get("sgpr_spill_count"), get("private_segment_fixed_size")


def check(files):
    bad = []
    with tempfile.TemporaryDirectory() as d:
        for f in files:
            cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-save-temps", "-c", os.path.join(CSRC, f),
                   "-o", os.path.join(d, f + ".o")] + FLAGS.get(f, [])
            subprocess.run(cmd, cwd=d, check=True, capture_output=True)
            asm = open(os.path.join(d, f.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
            n = 0
            for name, vgpr, vsp, ssp, scratch in kernels(asm):
                n += 1
                if vsp or scratch:
                    bad.append((f, name, vgpr, vsp, ssp, scratch))
                elif ssp:  # SGPR spills go to VGPR lanes, not to memory: reported, not fatal
                    print(f"note: {name}: {ssp} SGPRs spilled to VGPR lanes")
            print(f"{f}: {n} kernels checked")
    return bad


if __name__ == "__main__":
    bad = check(sys.argv[1:] or ["gemm.hip", "attention.hip"])
    for b in bad:
        print("SPILL/SCRATCH:", b)
    sys.exit(1 if bad else 0)
