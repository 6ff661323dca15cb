#!/usr/bin/env python3
"""Copies the summaries tools/collect_profiles_r04.sh left under gpurun_out/r04/ into profiles/ under their tracked names, stamping
the JSON ones with the build id the collection ran on (bench.py compares it with the build it runs)."""
import json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "r04"), os.path.join(ROOT, "profiles")
bid = open(os.path.join(SRC, "build_id.txt")).read().split()[-1]

COPY = {
    "bench_line.json": "r04_bench_line.json",
    "per_shape_f16_one_stream.txt": "r04_bench_per_shape_one_stream.txt",
    "per_shape_f16_split_streams.txt": "r04_bench_per_shape_split_streams.txt",
    "kernel_stats_f16_one_stream.csv": "r04_bench_kernel_stats_one_stream.csv",
    "rocprof_f16_one_stream.json": "r04_bench_under_rocprof_one_stream.json",
    "per_shape_bf16x3_one_stream.txt": "r04_x3_per_shape.txt",
    "kernel_stats_bf16x3_one_stream.csv": "r04_x3_kernel_stats.csv",
    "per_shape_bf16_one_stream.txt": "r04_bf16_per_shape_one_stream.txt",
    "per_shape_train_vae.txt": "r04_train_vae_per_shape.txt",
    "per_shape_train_diffusion.txt": "r04_train_diffusion_per_shape.txt",
    "step_traffic_by_kernel_f16.txt": "r04_pmc_step_traffic_by_kernel.txt",
    "step_traffic_by_kernel_bf16x3.txt": "r04_x3_pmc_step_traffic_by_kernel.txt",
    "step_sequence_f16_one_stream.txt": "r04_bench_step_sequence_one_stream.txt",
}
for a, b in COPY.items():
    if os.path.exists(os.path.join(SRC, a)):
        shutil.copy(os.path.join(SRC, a), os.path.join(DST, b))
        print("copied", b)

for a, b in (("step_traffic_f16.json", "r04_pmc_step_traffic.json"), ("step_traffic_bf16x3.json", "r04_x3_pmc_step_traffic.json")):
    d = json.load(open(os.path.join(SRC, a)))
    d["build_id"] = bid
    json.dump(d, open(os.path.join(DST, b), "w"), indent=1)
    print("wrote", b, d["bytes_per_step"])

M, KP, NP = 16384, 1408, 1408
ALG = {"f16": M * KP * 2 + 3 * NP * KP * 2 + M * NP * 2, "bf16x3": M * KP * 4 + 3 * NP * KP * 4 + M * NP * 4}
NAMES = {"f16": ("pmc_ffn_conv_f16.json", "r04_pmc_traffic_ffn_conv.json",
                  "conv_gemm_fat_kernel<F16,BIAS,taps-inner,11,shared rows> FFN causal conv [16384x4224]x[4224x1408] IEEE half (256x352 tile, "
                  "K-blocked operands, tap-inner K order, one staged copy of the rows per K-chunk)"),
         "bf16x3": ("pmc_ffn_conv_bf16x3.json", "r04_x3_pmc_traffic_ffn_conv.json",
                    "conv_gemm_big_kernel<BF16X3,BIAS> FFN causal conv [16384x4224]x[4224x1408] split operands (256x256 tile, K-tile pairs, term-outer)")}
for dt, (a, b, kernel) in NAMES.items():
    raw = json.load(open(os.path.join(SRC, a)))
    f, w = raw["FETCH_SIZE"]["mean_per_launch"], raw["WRITE_SIZE"]["mean_per_launch"]
    d = {"kernel": kernel,
         "command": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --kernel-trace --output-format csv -- python3 bench.py --no-split "
                    "--steps 30 [--dtype bf16x3] (default dtype f16; tools/collect_profiles_r04.sh); tools/pmc_summary.py <kernel>[@grid] <dirs>: mean per launch over "
                    f"{raw['FETCH_SIZE']['launches']} launches INSIDE the chain",
         "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
         "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM/rocprofv3); WRITE_SIZE exact; both in KiB",
         "hbm_bytes_per_launch": int((2 * f + w) * 1024),
         "algorithmic_bytes_per_launch": ALG[dt],
         "algorithmic_bytes_note": "activations [16384,1408] read once + 3 tap matrices [1408,1408] + output [16384,1408] written once"
                                   + (" (4 bytes per element: split rows)" if dt == "bf16x3" else " (2 bytes per element)"),
         "build_id": bid}
    json.dump(d, open(os.path.join(DST, b), "w"), indent=1)
    print("wrote", b, d["hbm_bytes_per_launch"])
# the weight-gradient kernel of the VAE training update (roofline.traffic of the train leg)
wp = os.path.join(SRC, "pmc_wgrad_vae.json")
if os.path.exists(wp):
    raw = json.load(open(wp))
    if "FETCH_SIZE" in raw and "WRITE_SIZE" in raw:
        f, w = raw["FETCH_SIZE"]["mean_per_launch"], raw["WRITE_SIZE"]["mean_per_launch"]
        d = {"kernel": "wgrad_tn_kernel<4>: weight gradient of the VAE's FFN causal conv k=3 (2048 x 6144 from ~12 k frames, 192 tiles of 256 x 256, one slice)",
             "command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace -- python3 bench.py --mode train --steps 6 --warmup 3; tools/pmc_summary.py wgrad_tn_kernel@98304x1",
             "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "correction": "gfx950: FETCH_SIZE x2; KiB", "hbm_bytes_per_launch": int((2 * f + w) * 1024),
             "launches": raw["FETCH_SIZE"]["launches"], "build_id": bid}
        json.dump(d, open(os.path.join(DST, "r04_pmc_train_wgrad_traffic.json"), "w"), indent=1)
        print("wrote r04_pmc_train_wgrad_traffic.json", d["hbm_bytes_per_launch"])
print("build", bid)
