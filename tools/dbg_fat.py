#!/usr/bin/env python3
"""Debug helper: compares the 256x352 tile against the 256x256 tile on one causal-conv problem and prints where they differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnorm_amd import _lib, ops, packing
cin, cout, k, dil, B, T = [int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (64, 352, 3, 2, 2, 300))]
dev = "cuda:0"
g = torch.Generator().manual_seed(11)
x = torch.randn(B, T, cin, generator=g)
w = torch.randn(cout, cin, k, generator=g) * (1.0 / (cin * k)) ** 0.5
xa = torch.zeros(B * T, packing.padk(cin)); xa[:, :cin] = x.view(B * T, cin)
xa = xa.to(dev, torch.bfloat16)
W = packing._conv(w, _lib.DN_BF16).to(dev)
bias = torch.zeros(W.shape[1], device=dev)
outs = {}
for tile in (3, 4):
    out = torch.full((B * T, cout), float("nan"), device=dev)
    ops.conv_gemm([(xa, W[j], (k - 1 - j) * dil) for j in range(k)], out, T, cout, bias=bias, tile=tile)
    torch.cuda.synchronize()
    outs[tile] = out.cpu()
d = (outs[3] - outs[4]).abs()
print("max diff", d.max().item(), "nan", torch.isnan(outs[4]).sum().item())
bad = (d > 1e-3)
rows = bad.any(1).nonzero().flatten()
cols = bad.any(0).nonzero().flatten()
print("bad rows", rows.numel(), rows[:40].tolist(), "...", rows[-10:].tolist())
print("bad cols", cols.numel(), cols[:40].tolist(), "...", cols[-10:].tolist())
