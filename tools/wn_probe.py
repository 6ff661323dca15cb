#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnorm_amd import _lib, ops, packing
dev = torch.device("cuda:0"); B, T = 32, 512; M = B * T; dt = torch.bfloat16
def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
G = 8
a3 = (torch.randn(G, M, 512, device=dev) * 0.5).to(dt)
a1 = (torch.randn(G, M, 1536, device=dev) * 0.5).to(dt)
w3 = [(torch.randn(G, 512, 512, device=dev) * 0.02).to(dt) for _ in range(3)]
w1 = (torch.randn(G, 512, 1536, device=dev) * 0.02).to(dt)
out = torch.empty(G, M, 512, device=dev, dtype=dt)
bias = torch.zeros(G, 512, device=dev)
res = torch.zeros_like(out); gb = torch.ones(1, G * 1024, device=dev)
fl = 2.0 * G * M * 1536 * 512
def rep(name, fn):
    s = timeit(fn); print(f"{name:44s} {s*1e6:8.1f} us {fl/s/1e12:8.1f} TF/s", flush=True)
rep("1 term K=1536, BIAS, groups=8", lambda: ops.conv_gemm([(a1, w1, 0)], out, T, 512, bias=bias, groups=G))
rep("3 terms shift 0, BIAS, groups=8", lambda: ops.conv_gemm([(a3, w3[j], 0) for j in range(3)], out, T, 512, bias=bias, groups=G))
rep("3 terms shift 2,1,0 (no group shift), BIAS", lambda: ops.conv_gemm([(a3, w3[j], 2 - j) for j in range(3)], out, T, 512, bias=bias, groups=G))
rep("3 terms dilated by group, BIAS", lambda: ops.conv_gemm([(a3, w3[j], 2 - j) for j in range(3)], out, T, 512, bias=bias, groups=G, shift_by_group=True))
rep("3 terms dilated, FILM_GATE", lambda: ops.conv_gemm([(a3, w3[j], 2 - j) for j in range(3)], out, T, 512, bias=bias, groups=G, shift_by_group=True,
    epilogue=_lib.EPI_FILM_GATE, res=res, gamma_beta=gb, gb_shared=True, gb_half=512))
rep("3 terms dilated, FILM_GATE, shared A (stack 0)", lambda: ops.conv_gemm([(a3[0], w3[j], 2 - j) for j in range(3)], out, T, 512, bias=bias, groups=G, shift_by_group=True,
    epilogue=_lib.EPI_FILM_GATE, res=res, gamma_beta=gb, gb_shared=True, gb_half=512, a_grouped=False))
for g in (1,):
    o1 = torch.empty(M, 512, device=dev, dtype=dt)
    s = timeit(lambda: ops.conv_gemm([(a1[0], w1[0], 0)], o1, T, 512, bias=bias[0]))
    print(f"single [16384x1536]x[1536x512] BIAS              {s*1e6:8.1f} us {fl/8/s/1e12:8.1f} TF/s")
