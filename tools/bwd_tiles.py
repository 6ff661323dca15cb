"""What would tap-inner order + one staged copy of the rows be worth for the BACKWARD-data form of a causal conv (negative shifts: it
runs term-outer today)?  Times the forward form (positive shifts: taps + shared rows on the 256-row tiles) and the backward-data form
of the FFN conv at the two training shapes on every tile.  python tools/bwd_tiles.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnorm_amd import ops, packing
dev = "cuda:0"


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for C, B in ((2048, 24), (1365, 16)):
    T = 512
    M = B * T
    Cp = packing.padk(C)
    x = (torch.randn(M, Cp, device=dev) * 0.5).bfloat16()
    wt = (torch.randn(3, packing.padn(C), Cp, device=dev) * 0.02).bfloat16()
    out = torch.empty(M, Cp, device=dev, dtype=torch.bfloat16)
    bias = torch.zeros(packing.padn(C), device=dev)
    for name, sign in (("forward (shifts 2,1,0)", 1), ("backward-data (shifts -2,-1,0)", -1)):
        for tile in (0, 1, 2, 3, 4):
            if tile == 4 and Cp % 352:
                continue
            try:
                t = timeit(lambda: ops.conv_gemm([(x, wt[j], sign * (2 - j)) for j in range(3)], out, T, Cp, bias=bias, tile=tile))
                print(f"FFN conv {C} M={M} {name} tile {tile}: {t:8.1f} us", flush=True)
            except Exception as e:
                print(f"FFN conv {C} M={M} {name} tile {tile}: {str(e)[:80]}", flush=True)
