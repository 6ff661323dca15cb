"""A/B of the guided step of the conditional eps-predictor on one box: per-step time of the captured chain (difference of
an 85- and a 22-step chain) with the library's run-time options switched through dn_set_option.  python tools/cond_ab.py [dtype]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffnorm_amd import _lib, engine, ops, scheduler, synthetic


def per_step(eng, x, lengths, prompt, plens, coef, n=21):
    out = []
    for steps in (n + 1, 4 * n + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev = eng.guided_ddim_chain(x, lengths, prompt, plens, steps, coef, cond_scale=2.0)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0, ev))
    return (out[1][0] - out[0][0]) / (out[1][1] - out[0][1]) * 1e3


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
    dev = torch.device("cuda", 0)
    B, T, Tp = 32, 512, 256
    cfg = synthetic.eps_config(dim_prompt=768, num_latents_m=64)
    eng = engine.EpsEngine(synthetic.random_eps_state_dict(cfg, seed=2), cfg, dtype=dtype, device=dev)
    coef = scheduler.DDPMScheduler(1000).ddim_coef_table(dev)
    lengths = torch.full((B,), T, dtype=torch.int32, device=dev)
    plens = torch.full((B,), Tp, dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream(device=dev)
    variants = [("default", {}), ("no_split_norm", {"no_split_norm": 1}), ("no_split_norm+kblock0", {"no_split_norm": 1, "kblock": 0}), ("kblock0", {"kblock": 0})]
    if os.environ.get("COND_AB_ONLY"):
        variants = [v for v in variants if v[0] == os.environ["COND_AB_ONLY"]]
    with torch.cuda.stream(stream):
        x = ops.randn((B, T, cfg.latent_dim), seed=77, device=dev)
        prompt = ops.randn((B, Tp, 768), seed=78, device=dev)
        eng.guided_ddim_chain(x, lengths, prompt, plens, 5, coef, cond_scale=2.0)
        for rep in range(3):
            for name, opts in variants:
                for k, v in opts.items():
                    _lib.set_option(k, v)
                x.copy_(ops.randn((B, T, cfg.latent_dim), seed=77, device=dev))
                ms = per_step(eng, x, lengths, prompt, plens, coef)
                for k in opts:
                    _lib.set_option(k, None)
                print(f"rep {rep} {name:24s} {ms:.3f} ms / guided step", flush=True)


if __name__ == "__main__":
    main()
