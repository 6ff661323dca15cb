#!/usr/bin/env python3
"""Golden vectors for the rest of GaussianDiffusion reachable through create_diffusion (diffusion/__init__.py:10-46): the START_X
mean type (predict_xstart=True), cond_fn guidance in p_sample / ddim_sample (condition_mean / condition_score, gaussian_diffusion.py:
346-374), ddim_sample_loop (:600-680), _prior_bpd and calc_bpd_loop (:788-858), incl. a respaced diffusion.  Outputs of the REAL
reference on a fixed closed-form "model" (the same one oracle/gen_golden.py uses for the scheduler fixtures) with every random draw
recorded.  Run in the build container only: python oracle/gen_golden_gd_tail.py"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402
from gen_golden import record_draws, save  # noqa: E402
from gen_golden_configs import seeded  # noqa: E402


def toy_model(C, learned):
    """A deterministic stand-in network: smooth in x, depends on t, 2C channels when the variance is learned."""
    def f(x, t, **kw):
        tt = t.float().view(-1, *([1] * (x.dim() - 1)))
        base = torch.tanh(0.7 * x + 0.01 * tt) - 0.1 * x
        return torch.cat([base, torch.sin(1.3 * x + 0.02 * tt)], dim=1) if learned else base
    return f


def cond_fn(x, t, **kw):  # grad log p(y | x) of a toy classifier
    tt = t.float().view(-1, *([1] * (x.dim() - 1)))
    return 0.3 * torch.cos(x) - 0.001 * tt


def denoised_fn(x):  # "a function which applies to the x_start prediction before it is used to sample" (:263-265); before the clip
    return 0.8 * torch.tanh(1.5 * x)


def main():
    torch.manual_seed(20261005)  # the recorded draws come off torch's global generator: a re-run regenerates the file byte for byte
    _, gd = ref_loader.load_reference()
    out = {}
    N, C, L = 3, 4, 10
    x = seeded((N, C, L), 501)
    for name, kw, learned in (("sx", dict(predict_xstart=True, learn_sigma=False), False),
                              ("sx_lr", dict(predict_xstart=True, learn_sigma=True), True),
                              ("eps", dict(learn_sigma=False, sigma_small=True), False)):
        diff = gd.create_diffusion(timestep_respacing="", diffusion_steps=100, **kw)
        model = toy_model(C, learned)
        t = torch.tensor([0, 37, 99])
        for clip in (False, True):
            tag = f"{name}_clip{int(clip)}"
            with record_draws() as rec:
                o = diff.p_sample(model, x, t, clip_denoised=clip)
            out[f"{tag}_p_noise"], out[f"{tag}_p_sample"], out[f"{tag}_p_x0"] = rec.draws[0], o["sample"], o["pred_xstart"]
            with record_draws() as rec:
                o = diff.p_sample(model, x, t, clip_denoised=clip, cond_fn=cond_fn, model_kwargs={})
            out[f"{tag}_pc_noise"], out[f"{tag}_pc_sample"] = rec.draws[0], o["sample"]
            with record_draws() as rec:
                o = diff.ddim_sample(model, x, t, clip_denoised=clip, cond_fn=cond_fn, model_kwargs={}, eta=0.5)
            out[f"{tag}_dc_noise"], out[f"{tag}_dc_sample"], out[f"{tag}_dc_x0"] = rec.draws[0], o["sample"], o["pred_xstart"]
        pm = diff.p_mean_variance(model, x, t, clip_denoised=True)
        out[f"{name}_pmv_mean"], out[f"{name}_pmv_logvar"] = pm["mean"], pm["log_variance"]
        # denoised_fn (:254-332 process_xstart; threaded through p_sample / ddim_sample / ddim_reverse_sample)
        for clip in (False, True):
            tag = f"{name}_dfn_clip{int(clip)}"
            pm = diff.p_mean_variance(model, x, t, clip_denoised=clip, denoised_fn=denoised_fn)
            out[f"{tag}_pmv_mean"], out[f"{tag}_pmv_x0"] = pm["mean"], pm["pred_xstart"]
            with record_draws() as rec:
                o = diff.p_sample(model, x, t, clip_denoised=clip, denoised_fn=denoised_fn, cond_fn=cond_fn, model_kwargs={})
            out[f"{tag}_p_noise"], out[f"{tag}_p_sample"], out[f"{tag}_p_x0"] = rec.draws[0], o["sample"], o["pred_xstart"]
            with record_draws() as rec:
                o = diff.ddim_sample(model, x, t, clip_denoised=clip, denoised_fn=denoised_fn, eta=0.5)
            out[f"{tag}_d_noise"], out[f"{tag}_d_sample"], out[f"{tag}_d_x0"] = rec.draws[0], o["sample"], o["pred_xstart"]
            o = diff.ddim_reverse_sample(model, x, t, clip_denoised=clip, denoised_fn=denoised_fn, eta=0.0)
            out[f"{tag}_r_sample"], out[f"{tag}_r_x0"] = o["sample"], o["pred_xstart"]
        with record_draws() as rec:
            y = diff.ddim_sample_loop(model, (N, C, L), noise=seeded((N, C, L), 504), clip_denoised=True, denoised_fn=denoised_fn, eta=0.3, device="cpu")
        out[f"{name}_dfn_dloop_noises"], out[f"{name}_dfn_dloop_out"] = torch.stack(rec.draws), y
        with record_draws() as rec:  # (training_losses unpacks `model_output, misc = model(...)`)
            tl = diff.training_losses(lambda *a, **k: (model(*a, **k), None), x, t)
        out[f"{name}_tl_noise"] = rec.draws[0]
        for k, v in tl.items():
            if torch.is_tensor(v):
                out[f"{name}_tl_{k}"] = v
        with record_draws() as rec:
            bp = diff.calc_bpd_loop(model, x, clip_denoised=True)
        out[f"{name}_bpd_noises"] = torch.stack(rec.draws)  # drawn for t = 99 .. 0
        for k, v in bp.items():
            out[f"{name}_bpd_{k}"] = v
        out[f"{name}_prior_bpd"] = diff._prior_bpd(x)
        with record_draws() as rec:
            y = diff.ddim_sample_loop(model, (N, C, L), noise=seeded((N, C, L), 502), clip_denoised=True, eta=0.3, device="cpu")
        out[f"{name}_dloop_noises"], out[f"{name}_dloop_out"] = torch.stack(rec.draws), y
    # respaced + cond_fn: both model and cond_fn must see the original step indices
    diff = gd.create_diffusion(timestep_respacing="5", diffusion_steps=100, learn_sigma=False)
    with record_draws() as rec:
        y = diff.ddim_sample_loop(toy_model(C, False), (N, C, L), noise=seeded((N, C, L), 503), clip_denoised=False, cond_fn=cond_fn,
                                  model_kwargs={}, eta=0.0, device="cpu")
    out["resp_dloop_noises"], out["resp_dloop_out"] = torch.stack(rec.draws), y
    out["x"] = x
    save("gaussian_tail", **out)


if __name__ == "__main__":
    main()
