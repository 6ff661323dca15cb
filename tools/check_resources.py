#!/usr/bin/env python3
"""Compiles the HIP sources for gfx950 with -save-temps and fails if any kernel touches scratch INSIDE A LOOP (a spill in a
hand-scheduled K loop is a slowdown and breaks its counted vmcnt waits: scratch loads share that counter) or keeps a stack object
other than register spills.  Spills that are stored before a kernel's loops and reloaded after them (epilogue constants carried
across the K loop of the 256-register tiles) are reported, not fatal.
Usage: check_resources.py [file.hip ...]   (default: gemm_bf16.hip gemm_f32.hip gemm_x3.hip attention.hip wgrad_tn.hip)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "diffnorm_amd", "csrc")
FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def kernels(asm_text):
    for blk in asm_text.split("  - .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
        yield name, get("vgpr_count"), get("vgpr_spill_count"), get("sgpr_spill_count"), get("private_segment_fixed_size")


def scratch_in_loops(asm_text, name):
    """Number of scratch_* instructions of kernel `name` that sit in a basic block LLVM marks as part of a loop."""
    start = asm_text.find("\n" + name + ":")
    if start < 0:
        return -1
    body = asm_text[start: asm_text.find("s_endpgm", start)]
    in_loop, n = False, 0
    for line in body.split("\n"):
        t = line.strip()
        if re.match(r"^\.?LBB\d+_\d+:", t) or t.startswith("; %bb."):
            in_loop = "Loop" in t  # "in Loop: Header=..." / "=>This Inner Loop Header"
        elif t.startswith("scratch_") and in_loop:
            n += 1
    return n


def check(files):
    """-> list of offending (file, kernel, vgprs, vgpr spills, sgpr spills, scratch bytes, scratch instructions inside loops)."""
    from concurrent.futures import ThreadPoolExecutor

    bad = []
    with tempfile.TemporaryDirectory() as d:
        def compile_one(f):
            sub = os.path.join(d, f + ".d")
            os.makedirs(sub)
            cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-save-temps", "-c", os.path.join(CSRC, f),
                   "-o", os.path.join(sub, f + ".o")] + FLAGS.get(f, [])
            subprocess.run(cmd, cwd=sub, check=True, capture_output=True)
            return f, open(os.path.join(sub, f.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()

        with ThreadPoolExecutor(max_workers=4) as pool:  # the translation units compile side by side (the suite's slowest test otherwise)
            for f, asm in pool.map(compile_one, files):
                n = 0
                for name, vgpr, vsp, ssp, scratch in kernels(asm):
                    n += 1
                    if vsp or scratch:
                        inside = scratch_in_loops(asm, name)
                        if inside != 0 or scratch > 4 * vsp + 64:  # scratch beyond the spill slots = a stack object (e.g. the parameter block)
                            bad.append((f, name, vgpr, vsp, ssp, scratch, inside))
                        else:
                            print(f"note: {name}: {vsp} VGPRs spilled around its loops (none inside), {scratch} B of scratch")
                    elif ssp:  # SGPR spills go to VGPR lanes, not to memory: reported, not fatal
                        print(f"note: {name}: {ssp} SGPRs spilled to VGPR lanes")
                print(f"{f}: {n} kernels checked")
    return bad


if __name__ == "__main__":
    bad = check(sys.argv[1:] or ["gemm_bf16.hip", "gemm_f32.hip", "gemm_x3.hip", "attention.hip", "wgrad_tn.hip"])
    for b in bad:
        print("SPILL/SCRATCH:", b)
    sys.exit(1 if bad else 0)
