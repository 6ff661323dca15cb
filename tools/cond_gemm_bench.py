"""The conditioning projection of a diffusion-training update, gb = cond [B, 2048] . W_c^T [2048 -> 57344] (fp32: 470 MB of weights
against B = 16 rows): time per forced tile variant.  Usage: python tools/cond_gemm_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnorm_amd import ops, _lib
dev = "cuda:0"
B, K, N = 16, 2048, 57344
a = torch.randn(B, K, device=dev)
w = torch.randn(N, K, device=dev) * 0.02
bias = torch.zeros(N, device=dev)
out = torch.empty(B, N, device=dev)
for tile in (0, 1, 2, 3):
    for _ in range(3):
        ops.conv_gemm([(a, w, 0)], out, B, N, bias=bias, tile=tile)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv_gemm([(a, w, 0)], out, B, N, bias=bias, tile=tile)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"tile {tile}: {ms*1e3:.1f} us  {N*K*4/ms/1e9:.2f} TB/s of weights")
ref = a @ w.t()
print("max err", (out - ref).abs().max().item())
