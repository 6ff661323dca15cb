"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by running the REAL reference
(loaded where it lies under /root/reference by oracle/ref_loader.py) on seeded inputs.

Run in the build container only:  ``python oracle/gen_golden.py [--full]``

Each fixture stores inputs' seeds/shapes and the reference's outputs (data only -- no
reference source).  Weights are NOT stored: they come from the portable deterministic
generators ``make_eps_state_dict`` / ``make_vae_state_dict`` in diffnorm_oracle.py and are
loaded into the reference modules with ``load_state_dict(strict=True)``, which also pins the
state-dict key layout.  Random draws inside the reference are reproduced by
recording every tensor it draws (torch.randn / randn_like / randint wrapped while it runs).
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import diffnorm_oracle as O  # noqa: E402
import ref_loader  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")

from gen_golden_configs import (CHAIN_EPS, CHAIN_VAE, FULL_EPS, FULL_VAE, TINY_EPS, TINY_EPS_COND,  # noqa: E402
                                ragged_lengths, seeded)


def ref_eps_model(lm, cfg, sd):
    m = lm.Model(cfg.dim, cfg.latent_dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads,
                 wavenet_layers=cfg.wavenet_layers, wavenet_stacks=cfg.wavenet_stacks,
                 dim_cond_mult=cfg.dim_cond_mult)
    full = dict(sd)
    full["pos_embed._float_tensor"] = torch.zeros(1)
    m.load_state_dict(full, strict=True)
    return m.eval()


def ref_vae(lm, cfg, sd):
    v = lm.SpeechVAEEncoderDecoder(cfg.dim, cfg.latent_dim)
    v.load_state_dict(sd, strict=True)
    return v.eval()


def save(name, **arrays):
    os.makedirs(GOLD, exist_ok=True)
    out = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()}
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print("wrote", name, {k: v.shape for k, v in out.items()})


def gen_schedules(lm, gd):
    out = {}
    for T in (200, 1000):
        s = lm.DDPMScheduler(T)
        for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                  "sqrt_one_minus_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                  "posterior_mean_coef1", "posterior_mean_coef2"):
            out[f"ddpm{T}_{k}"] = getattr(s, k)
    d = gd.create_diffusion(timestep_respacing="", learn_sigma=False)
    for k in ("betas", "alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
              "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
        out[f"linear1000_{k}"] = getattr(d, k)
    d50 = gd.create_diffusion(timestep_respacing="ddim50", learn_sigma=False)
    out["ddim50_timestep_map"] = np.array(d50.timestep_map)
    out["ddim50_betas"] = d50.betas
    d3 = gd.create_diffusion(timestep_respacing="10,15,20", learn_sigma=False, diffusion_steps=300)
    out["sec300_timestep_map"] = np.array(d3.timestep_map)
    save("schedules", **out)


def gen_gaussian_diffusion(gd):
    """q_sample / p_sample / ddim_sample / training_losses known answers with a closed-form model."""
    x0 = seeded((3, 4, 6), 11)
    noise = seeded((3, 4, 6), 12)
    noise2 = seeded((3, 4, 6), 13)
    t = torch.tensor([0, 417, 999])
    model = lambda x, ts, **kw: 0.3 * x - 0.01 * ts.float().view(-1, 1, 1) / 100 + 0.05
    model2 = lambda x, ts, **kw: torch.cat([model(x, ts), torch.tanh(x)], dim=1)
    out = {}
    for name, kw, mdl in (("large", dict(learn_sigma=False), model),
                          ("small", dict(learn_sigma=False, sigma_small=True), model),
                          ("learned", dict(learn_sigma=True), model2)):
        d = gd.create_diffusion(timestep_respacing="", **kw)
        out[f"{name}_q_sample"] = d.q_sample(x0, t, noise=noise)
        torch.manual_seed(5)
        ps = d.p_sample(mdl, x0, t)
        out[f"{name}_p_sample"] = ps["sample"]
        out[f"{name}_pred_xstart"] = ps["pred_xstart"]
        torch.manual_seed(5)
        out[f"{name}_p_sample_noclip"] = d.p_sample(mdl, x0, t, clip_denoised=False)["sample"]
        torch.manual_seed(5)
        out[f"{name}_ddim_eta0"] = d.ddim_sample(mdl, x0, t)["sample"]
        torch.manual_seed(5)
        out[f"{name}_ddim_eta05"] = d.ddim_sample(mdl, x0, t, eta=0.5)["sample"]
    torch.manual_seed(5)
    out["p_sample_noise"] = torch.randn_like(x0)  # what randn_like drew after manual_seed(5)
    d = gd.create_diffusion(timestep_respacing="ddim50", learn_sigma=False)
    t50 = torch.tensor([0, 20, 49])
    torch.manual_seed(5)
    out["ddim50_p_sample"] = d.p_sample(model, x0, t50)["sample"]
    out["ddim50_q_sample"] = d.q_sample(x0, t50, noise=noise)
    dl = gd.create_diffusion(timestep_respacing="", learn_sigma=False)
    out["train_mse"] = dl.training_losses(lambda x, ts, **kw: (model(x, ts), None), x0, t, noise=noise2)["mse"]
    # a short full reverse loop (respaced to 5 steps) with the noise sequence replayed
    d5 = gd.create_diffusion(timestep_respacing="5", learn_sigma=False)
    torch.manual_seed(9)
    xT = torch.randn(2, 4, 6)
    torch.manual_seed(10)
    out["loop5_out"] = d5.p_sample_loop(model, (2, 4, 6), noise=xT, device="cpu")
    torch.manual_seed(10)
    out["loop5_noises"] = torch.stack([torch.randn(2, 4, 6) for _ in range(5)])  # drawn at i=4,3,2,1,0
    out["loop5_xT"] = xT
    save("gaussian_diffusion", **out)


def gen_gaussian_moments(gd):
    """The rest of SURVEY 8 a15: q_posterior_mean_variance, p_mean_variance (three variance types, clipped and not),
    _predict_xstart_from_eps, ddim_reverse_sample, _vb_terms_bpd and training_losses with a learned variance (MSE + VB,
    rescaled, and the KL loss types) -- reference outputs on seeded inputs with closed-form models.  x_start has entries at and
    beyond +-0.999 so that every branch of the discretised decoder likelihood (diffusion_utils.py:66-88) is taken at t = 0."""
    x0 = seeded((4, 4, 6), 21).clamp(-1.2, 1.2) * 0.8
    x0[0, 0, :3] = torch.tensor([-1.0, 1.0, 0.9995])
    x0[0, 1, :2] = torch.tensor([-0.9995, 0.5])
    xt = seeded((4, 4, 6), 22)
    noise = seeded((4, 4, 6), 23)
    t = torch.tensor([0, 1, 417, 999])
    model = lambda x, ts, **kw: 0.3 * x - 0.01 * ts.float().view(-1, 1, 1) / 100 + 0.05
    model2 = lambda x, ts, **kw: torch.cat([model(x, ts), torch.tanh(x)], dim=1)
    out = dict(x0=x0, xt=xt, noise=noise, t=t)
    for name, kw, mdl in (("large", dict(learn_sigma=False), model), ("small", dict(learn_sigma=False, sigma_small=True), model),
                          ("learned", dict(learn_sigma=True), model2)):
        d = gd.create_diffusion(timestep_respacing="", **kw)
        for clip in (True, False):
            pm = d.p_mean_variance(mdl, xt, t, clip_denoised=clip)
            for k in ("mean", "variance", "log_variance", "pred_xstart"):
                out[f"{name}_pmv{int(clip)}_{k}"] = pm[k] + torch.zeros_like(xt)
        tr = torch.tensor([0, 1, 417, 998])  # alphas_cumprod_next at 999 is 0: covered too
        for tt, tag in ((tr, "a"), (t, "b")):
            rv = d.ddim_reverse_sample(mdl, xt, tt)
            out[f"{name}_reverse_{tag}"] = rv["sample"]
        vb = d._vb_terms_bpd(mdl, x0, xt, t, clip_denoised=False)
        out[f"{name}_vb_output"] = vb["output"]
    d = gd.create_diffusion(timestep_respacing="", learn_sigma=False)
    qm, qv, ql = d.q_posterior_mean_variance(x0, xt, t)
    out.update(qpost_mean=qm, qpost_var=qv + torch.zeros_like(xt), qpost_logvar=ql + torch.zeros_like(xt),
               xstart_from_eps=d._predict_xstart_from_eps(xt, t, noise))
    out["t_reverse_a"] = torch.tensor([0, 1, 417, 998])
    tl_model = lambda x, ts, **kw: (model2(x, ts), None)
    for name, kw in (("learned_mse", dict(learn_sigma=True)), ("learned_rescaled", dict(learn_sigma=True, rescale_learned_sigmas=True)),
                     ("learned_kl", dict(learn_sigma=True, use_kl=True))):
        d = gd.create_diffusion(timestep_respacing="", **kw)
        terms = d.training_losses(tl_model if "kl" not in name else model2, x0, t, noise=noise)
        for k in ("loss", "mse", "vb"):
            if k in terms:
                out[f"{name}_{k}"] = terms[k]
    d = gd.create_diffusion(timestep_respacing="ddim50", learn_sigma=True)  # respaced + learned: the wrapped model sees original steps
    t50 = torch.tensor([0, 1, 20, 49])
    terms = d.training_losses(tl_model, x0, t50, noise=noise)
    out.update(ddim50_learned_loss=terms["loss"], ddim50_learned_vb=terms["vb"], t50=t50)
    save("gaussian_moments", **out)


def gen_eps_tiny(lm):
    cfg = TINY_EPS
    sd = O.make_eps_state_dict(cfg, "tiny")
    m = ref_eps_model(lm, cfg, sd)
    B, T = 3, 40
    x = seeded((B, T, cfg.latent_dim), 21)
    lens = torch.tensor([40, 23, 31])
    mask = O.lengths_to_mask(lens, T)
    t = torch.tensor([3, 500, 999])
    with torch.no_grad():
        eps = m(x, t, input_mask=mask, cond_drop_prob=0)
        tc = m.to_time_cond(t)
        h = m.init_conv(x.transpose(1, 2))
        wn = m.wavenet(h, tc).transpose(1, 2)
        pe = m.pos_embed(mask)
    save("eps_tiny", x=x, t=t, lens=lens, eps=eps, time_cond=tc, wavenet=wn, pos_emb=pe)


def gen_eps_cond_tiny(lm):
    """Conditional eps-predictor (use_cond=True): Model(condition_on_prompt=True) with a ragged prompt -- the
    conditioned pass (cond_drop_prob 0), the null pass (1), classifier-free guidance at scale 2 (forward_with_cond_scale,
    latent_module.py:813-826), and the two conditioning products (pooled prompt condition, resampled prompt latents)."""
    cfg = TINY_EPS_COND
    sd = O.make_eps_state_dict(cfg, "cond")
    m = lm.Model(cfg.dim, cfg.latent_dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads, wavenet_layers=cfg.wavenet_layers,
                 wavenet_stacks=cfg.wavenet_stacks, dim_cond_mult=cfg.dim_cond_mult, condition_on_prompt=True, dim_prompt=cfg.dim_prompt,
                 num_latents_m=cfg.num_latents_m, resampler_depth=cfg.resampler_depth)
    full = dict(sd)
    full["pos_embed._float_tensor"] = torch.zeros(1)
    full["perceiver_resampler.embed_positions._float_tensor"] = torch.zeros(1)
    m.load_state_dict(full, strict=True)
    m.eval()
    B, T, Tp = 3, 40, 21
    x = seeded((B, T, cfg.latent_dim), 81)
    lens, plens = torch.tensor([40, 23, 31]), torch.tensor([21, 9, 14])
    mask, pmask = O.lengths_to_mask(lens, T), O.lengths_to_mask(plens, Tp)
    prompt = seeded((B, Tp, cfg.dim_prompt), 82)
    t = torch.tensor([3, 120, 77])
    with torch.no_grad():
        cond = m(x, t, prompt=prompt.clone(), prompt_mask=pmask, input_mask=mask, cond_drop_prob=0.0)
        null = m(x, t, prompt=prompt.clone(), prompt_mask=pmask, input_mask=mask, cond_drop_prob=1.0)
        cfg2 = m.forward_with_cond_scale(x, t, prompt=prompt.clone(), prompt_mask=pmask, input_mask=mask, cond_scale=2.0)
        masked = prompt.clone().masked_fill_(~pmask.unsqueeze(2), 0)
        pc = m.to_prompt_cond(masked)
        c = m.perceiver_resampler(masked, mask=pmask)
    save("eps_cond_tiny", x=x, t=t, lens=lens, plens=plens, prompt=prompt, eps_cond=cond, eps_null=null, eps_cfg2=cfg2, prompt_cond=pc,
         resampled=c)


def gen_eps_full(lm):
    cfg = FULL_EPS
    sd = O.make_eps_state_dict(cfg, "full")
    m = ref_eps_model(lm, cfg, sd)
    B, T = 8, 256
    x = seeded((B, T, cfg.latent_dim), 0)
    lens = ragged_lengths(B, T, 1, lo=128)
    mask = O.lengths_to_mask(lens, T)
    t = torch.full((B,), 500, dtype=torch.long)
    with torch.no_grad():
        eps = m(x, t, input_mask=mask, cond_drop_prob=0)
        tc = m.to_time_cond(t[:1])
    save("eps_full_cfg2", lens=lens, t=t, eps=eps, time_cond=tc, x_seed=0)


def gen_vae_full(lm):
    """BASELINE config 1: 64 utterances of [128,768]; slices + checksums of the reference outputs."""
    cfg = FULL_VAE
    sd = O.make_vae_state_dict(cfg, "full")
    v = ref_vae(lm, cfg, sd)
    B, T = 64, 128
    feat = seeded((B, T, cfg.dim), 0)
    lens = ragged_lengths(B, T, 2, lo=64)
    mask = O.lengths_to_mask(lens, T)
    post_noise = seeded((B, T, cfg.z), 3)
    with torch.no_grad():
        x = feat.transpose(1, 2)
        for w in v.encoder_wave:
            x = w(x)
        params = x.transpose(1, 2)
        z = O.posterior_sample(params, post_noise)  # reference draws its own noise; injected here
        recon, logits = v.decode_feature(z, mask)
    units = (logits.argmax(-1) - 4).to(torch.int16)
    top2 = logits.topk(2, dim=-1).values
    save("vae_full_cfg1", lens=lens, params_head=params[:2], params_sum=params.sum(dim=(1, 2)),
         recon_head=recon[:2, :, :96], recon_sum=recon.sum(dim=(1, 2)), logits_head=logits[:2, :16],
         units=units, margin=(top2[..., 0] - top2[..., 1]))


class record_draws:
    """Wraps torch.randn / randn_like / randint while the reference runs and records every tensor
    it draws, in call order (randn_like on a transposed view is not re-drawable from a seed)."""

    def __enter__(self):
        self.draws = []
        self._orig = {n: getattr(torch, n) for n in ("randn", "randn_like", "randint")}
        for n, f in self._orig.items():
            setattr(torch, n, self._wrap(f))
        return self

    def _wrap(self, f):
        def g(*a, **k):
            out = f(*a, **k)
            self.draws.append(out.detach().clone())
            return out
        return g

    def __exit__(self, *exc):
        for n, f in self._orig.items():
            setattr(torch, n, f)


def gen_chain(lm):
    """Short DDIM chains (start_step 1, 5 and 50, T=200) + training loss dict on a small VAE+eps pair."""
    ecfg, vcfg = CHAIN_EPS, CHAIN_VAE
    esd = O.make_eps_state_dict(ecfg, "chain")
    vsd = O.make_vae_state_dict(vcfg, "chain")
    vae = ref_vae(lm, vcfg, vsd)
    wrapper = types.SimpleNamespace(encoder=vae)
    ldm = lm.LatentDiscreteModel(wrapper, ecfg.dim, vcfg.z, timesteps=200)
    full = dict(esd)
    full["pos_embed._float_tensor"] = torch.zeros(1)
    ldm.model.load_state_dict(full, strict=True)
    ldm.eval()
    B, T = 3, 48
    feat = seeded((B, T, vcfg.dim), 31)
    lens = torch.tensor([48, 29, 40])
    mask = O.lengths_to_mask(lens, T)
    g = torch.Generator().manual_seed(32)
    units = torch.randint(4, 1004, (B, T), generator=g)
    units = units.masked_fill(~mask, 0)
    out = dict(lens=lens, units=units)
    for start in (1, 5, 50):
        torch.manual_seed(100 + start)
        with record_draws() as rec:
            toks, match, total, recon = ldm.ddim_sample(feat, input_mask=mask, ref_units=units - 4, start_step=start)
        # draw 0: posterior noise [B,z,T] (distributions.py:38); draw 1: start noise [B,T,z] (:1409);
        # then one unused randn_like per step (:1435)
        assert len(rec.draws) == 2 + max(1, start - 1), len(rec.draws)
        out[f"s{start}_post_noise"] = rec.draws[0].transpose(1, 2).contiguous()
        out[f"s{start}_start_noise"] = rec.draws[1]
        out[f"s{start}_recon"] = recon
        out[f"s{start}_match"] = match
        out[f"s{start}_total"] = total
        out[f"s{start}_units"] = torch.cat(toks)
    # training forward (multitask=True default): t, posterior noise, beta0 jitter, true noise
    torch.manual_seed(77)
    with torch.no_grad(), record_draws() as rec:
        ld = ldm(feat, units, tgt_mask=mask)
    assert len(rec.draws) == 4, len(rec.draws)
    out.update(train_times=rec.draws[0], train_post=rec.draws[1].transpose(1, 2).contiguous(),
               train_jitter=rec.draws[2].contiguous(), train_true=rec.draws[3].contiguous(),
               **{"train_" + k: v for k, v in ld.items()})
    # VAE criterion pieces: SpeechVAEEncoderDecoder.forward with the recorded posterior noise
    torch.manual_seed(78)
    with torch.no_grad(), record_draws() as rec:
        mse, logits, kl = vae(feat, units, mask)
    assert len(rec.draws) == 1
    out.update(vae_post=rec.draws[0].transpose(1, 2).contiguous(), vae_mse=mse, vae_kl=kl,
               vae_logits_head=logits[:, :8])
    save("chain_small", **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also regenerate the full-size fixtures (minutes of CPU)")
    args = ap.parse_args()
    torch.set_grad_enabled(True)
    lm, gd = ref_loader.load_reference()
    gen_schedules(lm, gd)
    gen_gaussian_diffusion(gd)
    gen_gaussian_moments(gd)
    gen_eps_tiny(lm)
    gen_eps_cond_tiny(lm)
    gen_chain(lm)
    if args.full:
        gen_eps_full(lm)
        gen_vae_full(lm)


if __name__ == "__main__":
    main()
