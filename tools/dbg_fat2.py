#!/usr/bin/env python3
"""Debug helper: repeats the tile-variant test sequence in-process and reports which launch differs and where."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnorm_amd import _lib, ops, packing
dev = "cuda:0"
def seeded(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed); return torch.randn(*shape, generator=g) * scale
shapes = [(192, 704, 3, 1, 3, 100), (64, 352, 3, 2, 2, 300), (128, 1056, 1, 1, 1, 515), (64, 352, 1, 1, 2, 130), (1408, 1408, 3, 1, 4, 512)]
def run(dtype, tile, cin, cout, k, dil, B, T):
    code = _lib.DN_BF16 if dtype == "bf16" else _lib.DN_F32
    x = seeded((B, T, cin), 11); w = seeded((cout, cin, k), 12, (1.0 / (cin * k)) ** 0.5); b = seeded((cout,), 13, 0.1)
    xa = torch.zeros(B * T, packing.padk(cin)); xa[:, :cin] = x.view(B * T, cin)
    xa = xa.to(dev, torch.bfloat16 if dtype == "bf16" else torch.float32)
    W = packing._conv(w, code).to(dev)
    out = torch.full((B * T, cout), float("nan"), device=dev)
    bias = packing._vec(b, W.shape[1]).to(dev)
    ops.conv_gemm([(xa, W[j], (k - 1 - j) * dil) for j in range(k)], out, T, cout, bias=bias, tile=tile)
    return out.cpu()
for rep in range(3):
    for dtype in ("f32", "bf16"):
        for tile in (1, 2, 3, 4):
            for sh in shapes:
                o = run(dtype, tile, *sh)
                if tile == 1: ref = {**globals().get("ref", {}), (dtype, sh): o}; globals()["ref"] = ref
                else:
                    d = (o - ref[(dtype, sh)]).abs()
                    if not (d.max().item() < 1e-3):
                        bad = d > 1e-3
                        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
                        print("MISMATCH rep", rep, dtype, "tile", tile, sh, "max", d.max().item(), "nan", torch.isnan(o).sum().item(),
                              "rows", rows.numel(), rows[:12].tolist(), rows[-4:].tolist(), "cols", cols.numel(), cols[:8].tolist(), cols[-4:].tolist())
print("done")
