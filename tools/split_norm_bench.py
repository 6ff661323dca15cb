#!/usr/bin/env python3
"""Cost of the split-RMSNorm epilogue options on the step's shapes ([32,512] x dim 512): consumer (q/kv, GEGLU) and producer
(attention out, FFN out) contractions with and without them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnorm_amd import _lib, ops, packing

dev = torch.device("cuda:0")
B, T, D = 32, 512, 512
M = B * T
bf = torch.bfloat16

def timeit(fn, iters=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

xg = (torch.randn(M, D, device=dev) * 0.5).to(bf)
ssq = torch.rand(M, 8, device=dev) + 1.0
for name, N, epi in (("qkv 512->1536", 1536, _lib.EPI_BIAS), ("GEGLU 512->2x1408", 1408, _lib.EPI_GEGLU)):
    rows = N if epi == _lib.EPI_BIAS else 2 * N
    W = (torch.randn(rows, D, device=dev) * 0.02).to(bf)
    out = torch.empty(M, N, device=dev, dtype=bf)
    bias = torch.zeros(rows, device=dev)
    rb = torch.randn(1, rows, device=dev)
    rbB = torch.randn(B, rows, device=dev)
    t0 = timeit(lambda: ops.conv_gemm([(xg, W, 0)], out, T, N, bias=bias, epilogue=epi))
    t1 = timeit(lambda: ops.conv_gemm([(xg, W, 0)], out, T, N, bias=bias, epilogue=epi, row_ssq=ssq, row_D=D))
    t2 = timeit(lambda: ops.conv_gemm([(xg, W, 0)], out, T, N, bias=bias, epilogue=epi, row_ssq=ssq, row_D=D, row_bias=rb, row_bias_shared=True))
    t3 = timeit(lambda: ops.conv_gemm([(xg, W, 0)], out, T, N, bias=bias, epilogue=epi, row_ssq=ssq, row_D=D, row_bias=rbB))
    print(f"{name:22s} plain {t0:6.1f} us | +row scale {t1:6.1f} | +shared beta.W {t2:6.1f} | per-sample beta.W {t3:6.1f}")
for name, K in (("attn_out 512->512", 512), ("ffn_out 1408->512", 1408)):
    a = (torch.randn(M, K, device=dev) * 0.5).to(bf)
    W = (torch.randn(D, K, device=dev) * 0.02).to(bf)
    xres = torch.randn(M, D, device=dev)
    xn = torch.empty(M, D, device=dev, dtype=bf)
    sq = torch.empty(M, 8, device=dev)
    gb = torch.randn(1, 2 * D, device=dev)
    t0 = timeit(lambda: ops.conv_gemm([(a, W, 0)], xres, T, D, epilogue=_lib.EPI_RESADD, res=xres))
    t1 = timeit(lambda: ops.conv_gemm([(a, W, 0)], xres, T, D, epilogue=_lib.EPI_RESADD, res=xres, norm_out=xn, norm_D=D, norm_ssq=sq,
                                      norm_gb=gb, norm_gb_half=D, norm_gb_shared=True))
    xn2 = torch.empty(M, D, device=dev, dtype=bf)
    t2 = timeit(lambda: ops.rmsnorm(xres, xn2, T, gamma_beta=gb, gb_shared=True, gb_half=D))
    print(f"{name:22s} plain {t0:6.1f} us | +split-norm outputs {t1:6.1f} | standalone rmsnorm kernel {t2:6.1f}")
# consumer variants on the q/kv shape: partials fetched at kernel start (ld 8) vs in the epilogue (ld 9 -> not 16-byte rows), forced tiles
W = (torch.randn(1536, D, device=dev) * 0.02).to(bf)
out = torch.empty(M, 1536, device=dev, dtype=bf)
bias = torch.zeros(1536, device=dev)
ssq9 = torch.rand(M, 9, device=dev) + 1.0
for tile in (0, 1, 2, 3):
    t0 = timeit(lambda: ops.conv_gemm([(xg, W, 0)], out, T, 1536, bias=bias, tile=tile))
    t1 = timeit(lambda: ops.conv_gemm([(xg, W, 0)], out, T, 1536, bias=bias, tile=tile, row_ssq=ssq, row_D=D))
    t2 = timeit(lambda: ops.conv_gemm([(xg, W, 0)], out, T, 1536, bias=bias, tile=tile, row_ssq=ssq9, row_D=D))
    print(f"qkv tile {tile}: plain {t0:6.1f} us | row scale (requested at kernel start) {t1:6.1f} | row scale (fetched in the epilogue) {t2:6.1f}")
