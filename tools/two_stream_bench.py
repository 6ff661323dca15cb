#!/usr/bin/env python3
"""Experiment: one [32,512] chain vs two independent [16,512] chains on two streams (fill/drain overlap)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnorm_amd import engine, ops, scheduler, synthetic

dev = torch.device("cuda:0")
cfg = synthetic.eps_config()
sd = synthetic.random_eps_state_dict(cfg, seed=0)
K, W, T = 40, 5, 512
sched = scheduler.DDPMScheduler(1000)
coef = sched.ddim_coef_table(dev)

def run(nsplit):
    B = 32 // nsplit
    engs = [engine.EpsEngine(sd, cfg, dtype="bf16", device=dev) for _ in range(nsplit)]
    xs = [ops.randn((B, T, 128), seed=7 + i, device=dev) for i in range(nsplit)]
    lens = [torch.full((B,), T, dtype=torch.int32, device=dev) for _ in range(nsplit)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nsplit)]
    def go(start, n):
        for e, x, l, s in zip(engs, xs, lens, streams):
            with torch.cuda.stream(s):
                e.ddim_loop(x, l, start, coef, use_graph=True, max_evals=n)
    go(999, W)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go(999 - W, K)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"split {nsplit}: {dt / K * 1e3:.3f} ms per 32-sequence step")

for n in (1, 2, 1, 2, 4):
    run(n)
