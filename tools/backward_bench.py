#!/usr/bin/env python3
"""the two backward contractions of the FFN causal conv (k=3, 1365 -> 1365, [B=32,T=512]) through dn_conv_gemm --
data gradient (negative shifts, transposed weights) and weight gradient (ops.conv_weight_grad: transposes + split-K)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnorm_amd import _lib, ops, packing

dev = "cuda:0"
B, T, C = 32, 512, 1365
M, Cp = B * T, packing.padk(C)


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


x = (torch.randn(M, Cp, device=dev) * 0.5).to(torch.bfloat16)
dy = (torch.randn(M, Cp, device=dev) * 0.5).to(torch.bfloat16)
wt = (torch.randn(3, packing.padn(C), Cp, device=dev) * 0.02).to(torch.bfloat16)  # W_j^T packed
dx = torch.empty(M, Cp, device=dev, dtype=torch.bfloat16)
fl = 2.0 * M * C * 3 * C
t = timeit(lambda: ops.conv_gemm([(dy, wt[j], -(2 - j)) for j in range(3)], dx, T, Cp))
print(f"FFN conv dX (3 taps, negative shifts): {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TFLOP/s")
t = timeit(lambda: ops.conv_weight_grad(x, dy, T, C, C, [2, 1, 0]))
print(f"FFN conv dW (3 x transpose + split-K contraction + partial sums): {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TFLOP/s")
