#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnorm_amd import ops
dev = torch.device("cuda:0"); B, T, hd = 32, 512, 512; M = B * T
qkv = (torch.randn(M, 3 * hd, device=dev) * 0.5).to(torch.bfloat16)
ao = torch.empty(M, hd, device=dev, dtype=torch.bfloat16)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
for _ in range(10):
    ops.attention(qkv, qkv[:, hd:], qkv[:, 2 * hd:], ao, B, T, 8, 64, lens, ldq=3 * hd, ldk=3 * hd, ldv=3 * hd)
torch.cuda.synchronize()
