#!/usr/bin/env python3
"""Reference point only (not product code): what the vendor GEMM reaches on the step's contraction shapes."""
import torch
dev = torch.device("cuda:0")
def t(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
for name, M, K, N in (("ffn_conv", 16384, 4224, 1408), ("wn_dilated(1 of 8)", 16384, 1536, 512), ("ffn_in", 16384, 512, 2816),
                      ("qkv", 16384, 512, 1536), ("attn_out", 16384, 512, 512), ("cube4k", 4096, 4096, 4096), ("cube8k", 8192, 8192, 8192)):
    a = (torch.randn(M, K, device=dev) * 0.5).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    s = t(lambda: torch.matmul(a, w.t()))
    print(f"{name:20s} {s*1e6:8.1f} us {2.0*M*K*N/s/1e12:8.1f} TF/s")
a = (torch.randn(8, 16384, 1536, device=dev) * 0.5).bfloat16(); w = (torch.randn(8, 512, 1536, device=dev) * 0.02).bfloat16()
s = t(lambda: torch.bmm(a, w.transpose(1, 2)))
print(f"{'wn_dilated bmm x8':20s} {s*1e6:8.1f} us {2.0*8*16384*1536*512/s/1e12:8.1f} TF/s")
