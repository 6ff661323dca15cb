#!/usr/bin/env python3
"""HBM rate of the optimizer step: gradient-norm pass (4 B/element) and Adam update (28 B/element, 30 with the
bf16 working copy) over a flat buffer of the eps-predictor's size class; HIP events around `iters` back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnorm_amd import optim

dev = "cuda:0"
for n in (64 * 2**20, 384 * 2**20):  # 64 Mi and 384 Mi parameters (the eps-predictor has ~0.36 G)
    p = torch.randn(n, device=dev)
    g = torch.randn(n, device=dev) * 1e-3
    for shadow in (None, torch.empty(n, device=dev, dtype=torch.bfloat16)):
        opt = optim.Adam(p, lr=3e-4, betas=(0.9, 0.98), clip_norm=2.0, bf16_copy=shadow)
        def timed(fn, iters=10):
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters * 1e-3
        t_norm = timed(lambda: opt.grad_sumsq(g))
        t_all = timed(lambda: opt.step(g))
        bytes_upd = n * (28 + (2 if shadow is not None else 0))
        t_upd = t_all - t_norm
        print(f"n = {n/2**20:.0f} Mi, bf16 copy {'yes' if shadow is not None else 'no '}: norm {t_norm*1e3:7.3f} ms = {4*n/t_norm/1e12:5.2f} TB/s; "
              f"update {t_upd*1e3:7.3f} ms = {bytes_upd/t_upd/1e12:5.2f} TB/s; norm+update {t_all*1e3:7.3f} ms", flush=True)
        del opt
