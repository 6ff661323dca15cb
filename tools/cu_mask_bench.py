#!/usr/bin/env python3
"""Experiment: two independent [16,512] chains on two streams, with and without disjoint CU masks
(hipExtStreamCreateWithCUMask).  The question: do a chain's memory-bound phases (epilogue write bursts, residual-stream
passes) overlap the other chain's contractions better when each chain owns half of the CUs?"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnorm_amd import engine, ops, scheduler, synthetic

dev = torch.device("cuda:0")
cfg = synthetic.eps_config()
sd = synthetic.random_eps_state_dict(cfg, seed=0)
K, W, T = 40, 5, 512
sched = scheduler.DDPMScheduler(1000)
coef = sched.ddim_coef_table(dev)
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    err = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    assert err == 0, err
    return torch.cuda.ExternalStream(s.value, device=dev)


def run(name, streams, use_graph):
    n = len(streams)
    B = 32 // n
    engs = [engine.EpsEngine(sd, cfg, dtype="bf16", device=dev) for _ in range(n)]
    xs = [ops.randn((B, T, 128), seed=7 + i, device=dev) for i in range(n)]
    lens = [torch.full((B,), T, dtype=torch.int32, device=dev) for _ in range(n)]

    def go(start, k):
        for e, x, l, s in zip(engs, xs, lens, streams):
            with torch.cuda.stream(s):
                e.ddim_loop(x, l, start, coef, use_graph=use_graph, max_evals=k, split=False)

    go(999, W)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go(999 - W, K)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:44s} graph={int(use_graph)}: {dt / K * 1e3:.3f} ms per 32-sequence step", flush=True)


plain = lambda: [torch.cuda.Stream(device=dev) for _ in range(2)]
run("one stream [32,512]", [torch.cuda.Stream(device=dev)], True)
run("one stream [32,512]", [torch.cuda.Stream(device=dev)], False)
run("two plain streams", plain(), True)
run("two plain streams", plain(), False)
even, odd = [0x55555555] * 8, [0xAAAAAAAA] * 8
run("two streams, CU masks even / odd bits", [masked_stream(even), masked_stream(odd)], False)
lo, hi = [0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 4
run("two streams, CU masks low / high 128 bits", [masked_stream(lo), masked_stream(hi)], False)
b16 = [0x0000FFFF] * 8
t16 = [0xFFFF0000] * 8
run("two streams, CU masks 16-bit halves per word", [masked_stream(b16), masked_stream(t16)], False)
run("two plain streams", plain(), False)
