#!/usr/bin/env python3
"""BASELINE config 5 end to end on one GPU: [16,1024,768] features -> VAE encode -> noise at start_step -> full DDIM chain
(start_step - 1 evaluations) -> VAE decode -> units, once in bf16 and once in exact-fp32 arithmetic (the mode that matches the
reference's golden outputs to 5e-6 per evaluation), and how far the two agree after the whole chain.
Usage: full_chain_agreement.py [start_step=999] [B=16] [T=1024]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnorm_amd import engine, ops, scheduler, synthetic

dev = torch.device("cuda:0")
start = int(sys.argv[1]) if len(sys.argv) > 1 else 999
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
cfg = synthetic.eps_config()
esd = synthetic.random_eps_state_dict(cfg, seed=0)
vsd = synthetic.random_vae_state_dict(seed=1)
sched = scheduler.DDPMScheduler(1000)
coef = sched.ddim_coef_table(dev)
sa, s1 = sched.f32("sqrt_alphas_cumprod", dev), sched.f32("sqrt_one_minus_alphas_cumprod", dev)
g = torch.Generator().manual_seed(0)
feat = torch.randn(B, T, 768, generator=g)
lens = torch.randint(T // 2, T + 1, (B,), generator=g)
lens[0] = T
post = torch.randn(B, T, 128, generator=g)
noise = torch.randn(B, T, 128, generator=g)
res = {}
for dtype in ("f32", "bf16"):
    eps = engine.EpsEngine(esd, cfg, dtype=dtype, device=dev)
    vae = engine.VaeEngine(vsd, dtype=dtype, device=dev)
    z = vae.sample_posterior(vae.encode_params(feat.to(dev)), post)
    ts = torch.full((B,), start, dtype=torch.int32, device=dev)
    x = ops.q_sample(z, noise.to(dev), sa, s1, ts, T)
    st = torch.cuda.Stream()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(st):
        n = eps.ddim_loop(x, lens.to(dev).int(), start, coef)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    recon, logits, units = vae.decode(x, lens)
    res[dtype] = (x.cpu(), recon.cpu(), units.cpu())
    print(f"{dtype}: {n} evaluations in {dt:.2f} s ({n/dt:.1f} steps/s at [{B},{T}]), latent finite: {bool(torch.isfinite(x).all())}", flush=True)
    del eps, vae
    torch.cuda.empty_cache()
mask = torch.arange(T).view(1, -1) < lens.view(-1, 1)
xa, xb = res["f32"][0], res["bf16"][0]
ua, ub = res["f32"][2], res["bf16"][2]
print(f"after {start - 1} evaluations: latent rms {xa[mask].pow(2).mean().sqrt():.4f}, bf16-vs-f32 latent rms diff {(xa - xb)[mask].pow(2).mean().sqrt():.4e}, "
      f"recon rms diff {(res['f32'][1] - res['bf16'][1])[mask].pow(2).mean().sqrt():.4e}, unit agreement {(ua[mask] == ub[mask]).float().mean().item():.4f}")
