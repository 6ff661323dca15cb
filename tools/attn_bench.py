#!/usr/bin/env python3
"""Attention forward at constant B*T = 16384 rows over several T: separates the per-key-tile cost (grows with T) from the fixed
part (launch, prologue, epilogue).  Prints us per launch, PFLOP/s and cycles per (wave, 64-key tile) at the nominal 2.4 GHz."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnorm_amd import ops

dev = torch.device("cuda:0")
heads, dh = 8, 64
hd = heads * dh
for T in (128, 256, 512, 1024, 2048, 4096):
    B = 16384 // T
    M = B * T
    qkv = (torch.randn(M, 3 * hd, device=dev) * 0.5).to(torch.bfloat16)
    ao = torch.empty(M, hd, device=dev, dtype=torch.bfloat16)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    run = lambda: ops.attention(qkv, qkv[:, hd:], qkv[:, 2 * hd:], ao, B, T, heads, dh, lens, ldq=3 * hd, ldk=3 * hd, ldv=3 * hd)
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 100
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    flop = 4.0 * T * T * dh * heads * B
    wave_tiles_per_simd = (M // 32) * heads * (T // 64) / 1024
    print(f"T={T:5d} B={B:4d}: {us:8.1f} us  {flop / us * 1e-9:6.3f} PFLOP/s  {us * 2400 / wave_tiles_per_simd:7.0f} cycles per wave-tile per SIMD")
