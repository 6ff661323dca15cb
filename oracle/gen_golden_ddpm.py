#!/usr/bin/env python3
"""Golden vectors for the ancestral (DDPM) device loop, dn_ddpm_loop: the REAL reference's GaussianDiffusion.p_sample
(diffusion/gaussian_diffusion.py:376-417) driven step by step over the REAL reference eps-predictor (latent_module.Model, the
chain-sized config) on the cosine schedule of DDPMScheduler (T = 200), five steps t = 4 .. 0 from a noised latent, with every
torch.randn_like the reference draws recorded (the injected noise of the parity run).  Both fixed variances create_diffusion can
build (FIXED_SMALL: learn_sigma=False; FIXED_LARGE: sigma_small=False) and clip_denoised on / off.
Run in the build container only: python oracle/gen_golden_ddpm.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import diffnorm_oracle as O  # noqa: E402
import ref_loader  # noqa: E402
from gen_golden import record_draws, ref_eps_model, save  # noqa: E402
from gen_golden_configs import CHAIN_EPS, seeded  # noqa: E402


def main():
    lm, gd = ref_loader.load_reference()
    G = sys.modules["refdiffusion.gaussian_diffusion"]
    cfg = CHAIN_EPS
    model = ref_eps_model(lm, cfg, O.make_eps_state_dict(cfg, "chain"))
    B, T, start = 3, 48, 5
    lens = torch.tensor([48, 29, 40])
    mask = O.lengths_to_mask(lens, T)
    betas = lm.get_named_beta_schedule("cosine", 200)
    out = dict(lens=lens)
    z0 = seeded((B, T, cfg.latent_dim), 71)
    for name, var_type, clip in (("small", G.ModelVarType.FIXED_SMALL, False), ("large", G.ModelVarType.FIXED_LARGE, False),
                                 ("small_clip", G.ModelVarType.FIXED_SMALL, True)):
        diff = G.GaussianDiffusion(betas=betas, model_mean_type=G.ModelMeanType.EPSILON, model_var_type=var_type, loss_type=G.LossType.MSE)
        fn = lambda x, t: model(x, t, input_mask=mask, cond_drop_prob=0)  # noqa: E731  (elementwise scheduler: the [B,T,z] layout is fine)
        torch.manual_seed(300)
        x = diff.q_sample(z0, torch.full((B,), start - 1, dtype=torch.long), noise=seeded((B, T, cfg.latent_dim), 72))
        out[f"{name}_x_start"] = x.clone()
        noises = []
        with torch.no_grad():
            for t in range(start - 1, -1, -1):
                with record_draws() as rec:
                    x = diff.p_sample(fn, x, torch.full((B,), t, dtype=torch.long), clip_denoised=clip)["sample"]
                assert len(rec.draws) == 1 and rec.draws[0].shape == x.shape
                noises.append(rec.draws[0])
        out[f"{name}_noise"] = torch.stack(noises)  # row k belongs to step t = start-1-k
        out[f"{name}_x_end"] = x
    save("ddpm_chain", **out)


if __name__ == "__main__":
    main()
