#!/usr/bin/env python3
"""Weight gradient of the FFN causal convs at training sizes: transposed copies + dn_conv_gemm (ops.conv_weight_grad) against the
row-major form (ops.conv_weight_grad_tn, transposing LDS reads), HIP-event times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnorm_amd import ops

dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for name, cin, cout, shifts, B, T in (("eps ffn conv 1365 k3", 1365, 1365, [2, 1, 0], 20, 400), ("vae ffn conv 2048 k3", 2048, 2048, [2, 1, 0], 30, 400),
                                      ("ffn_in 512->2730", 512, 2816, [0], 20, 400), ("ffn_out 1365->512", 1365, 512, [0], 20, 400),
                                      ("qkv 512->1536", 512, 1536, [0], 20, 400)):
    pk = lambda c: (c + 63) // 64 * 64
    x = (torch.randn(B * T, pk(cin), device=dev) * 0.5).bfloat16()
    dy = (torch.randn(B * T, pk(cout), device=dev) * 0.5).bfloat16()
    x[:, cin:] = 0
    dy[:, cout:] = 0
    flops = 2.0 * B * T * cin * cout * len(shifts)
    t_nt = timeit(lambda: ops.conv_weight_grad(x, dy, T, cin, cout, shifts))
    line = f"{name:24s} transposes+NT {t_nt:8.1f} us ({flops / t_nt / 1e6:6.1f} TF/s)"
    for sl in (1, 2, 4):
        t = timeit(lambda: ops.conv_weight_grad_tn(x, dy, T, cin, cout, shifts, slices=sl))
        line += f" | TN s{sl} {t:8.1f} us ({flops / t / 1e6:6.1f})"
    a, b = ops.conv_weight_grad(x, dy, T, cin, cout, shifts), ops.conv_weight_grad_tn(x, dy, T, cin, cout, shifts, slices=2)
    line += f" | max diff {(a - b).abs().max().item():.2e} of {a.abs().max().item():.2e}"
    print(line, flush=True)
