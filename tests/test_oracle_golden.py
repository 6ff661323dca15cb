"""Pins the CPU oracle (oracle/diffnorm_oracle.py) to outputs of the real reference.

The fixtures under tests/golden/ were produced by oracle/gen_golden.py, which runs the
reference itself (build container only).  These tests need neither the reference nor a GPU.
Tolerances: fp32 round-off only (the oracle issues the same torch ops in the same order).
"""
import numpy as np
import pytest
import torch

import diffnorm_oracle as O
from gen_golden_configs import CHAIN_EPS, CHAIN_VAE, FULL_EPS, FULL_VAE, TINY_EPS, ragged_lengths, seeded


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, tol):
    a, b = T(a).double(), T(b).double()
    err = (a - b).abs().max().item()
    assert err <= tol, f"max abs err {err} > {tol}"


def test_schedule_tables(golden):
    g = golden("schedules")
    for n in (200, 1000):
        tab = O.ddpm_tables(n)
        for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                  "sqrt_one_minus_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                  "posterior_mean_coef1", "posterior_mean_coef2"):
            np.testing.assert_allclose(getattr(tab, k), g[f"ddpm{n}_{k}"], rtol=1e-14, atol=0)
    # known answers quoted in SURVEY.md 8(a10)
    assert abs(O.ddpm_tables(200).betas[0] - 2.549726363721e-4) < 1e-15
    assert abs(O.ddpm_tables(1000).alphas_cumprod[500] - 0.4922851724488) < 1e-12
    lin = O.create_diffusion_oracle("", learn_sigma=False)
    for k in ("betas", "alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
              "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
        np.testing.assert_allclose(getattr(lin.tab, k), g[f"linear1000_{k}"], rtol=1e-14, atol=0)
    d50 = O.create_diffusion_oracle("ddim50", learn_sigma=False)
    assert d50.timestep_map == g["ddim50_timestep_map"].tolist()
    np.testing.assert_allclose(d50.tab.betas, g["ddim50_betas"], rtol=1e-13)
    d3 = O.create_diffusion_oracle("10,15,20", learn_sigma=False, diffusion_steps=300)
    assert d3.timestep_map == g["sec300_timestep_map"].tolist()


def _toy(x, ts):
    return 0.3 * x - 0.01 * ts.float().view(-1, 1, 1) / 100 + 0.05


def _toy2(x, ts):
    return torch.cat([_toy(x, ts), torch.tanh(x)], dim=1)


def test_gaussian_diffusion(golden):
    g = golden("gaussian_diffusion")
    x0, noise, noise2 = seeded((3, 4, 6), 11), seeded((3, 4, 6), 12), seeded((3, 4, 6), 13)
    t = torch.tensor([0, 417, 999])
    pn = T(g["p_sample_noise"])
    for name, kw, mdl in (("large", dict(learn_sigma=False), _toy),
                          ("small", dict(learn_sigma=False, sigma_small=True), _toy),
                          ("learned", dict(learn_sigma=True), _toy2)):
        d = O.create_diffusion_oracle("", **kw)
        close(d.q_sample(x0, t, noise), g[f"{name}_q_sample"], 1e-6)
        ps = d.p_sample(mdl, x0, t, pn)
        close(ps["sample"], g[f"{name}_p_sample"], 2e-6)
        close(ps["pred_xstart"], g[f"{name}_pred_xstart"], 2e-6)
        close(d.p_sample(mdl, x0, t, pn, clip_denoised=False)["sample"], g[f"{name}_p_sample_noclip"], 2e-5)
        close(d.ddim_sample(mdl, x0, t, pn)["sample"], g[f"{name}_ddim_eta0"], 2e-6)
        close(d.ddim_sample(mdl, x0, t, pn, eta=0.5)["sample"], g[f"{name}_ddim_eta05"], 2e-6)
    d = O.create_diffusion_oracle("ddim50", learn_sigma=False)
    t50 = torch.tensor([0, 20, 49])
    close(d.p_sample(_toy, x0, t50, pn)["sample"], g["ddim50_p_sample"], 2e-6)
    close(d.q_sample(x0, t50, noise), g["ddim50_q_sample"], 1e-6)
    close(O.create_diffusion_oracle("", learn_sigma=False).training_mse(_toy, x0, t, noise2), g["train_mse"], 1e-6)
    d5 = O.create_diffusion_oracle("5", learn_sigma=False)
    noises = T(g["loop5_noises"])  # drawn for i = 4,3,2,1,0 in that order
    by_step = {i: noises[4 - i] for i in range(5)}
    close(d5.p_sample_loop(_toy, T(g["loop5_xT"]), by_step), g["loop5_out"], 5e-6)


def test_gaussian_moments(golden):
    """The rest of SURVEY 8 a15 (posterior / model moments, DDIM reverse step, VB terms, learned-variance training losses)."""
    g = golden("gaussian_moments")
    x0, xt, noise, t = T(g["x0"]), T(g["xt"]), T(g["noise"]), T(g["t"])
    for name, kw, mdl in (("large", dict(learn_sigma=False), _toy), ("small", dict(learn_sigma=False, sigma_small=True), _toy),
                          ("learned", dict(learn_sigma=True), _toy2)):
        d = O.create_diffusion_oracle("", **kw)
        for clip in (True, False):
            pm = d.p_mean_variance(mdl, xt, t, clip_denoised=clip)
            for k in ("mean", "variance", "log_variance", "pred_xstart"):
                ref = g[f"{name}_pmv{int(clip)}_{k}"]
                close(pm[k], ref, 2e-6 * max(1.0, float(np.abs(ref).max())))
        close(d.ddim_reverse_sample(mdl, xt, T(g["t_reverse_a"]))["sample"], g[f"{name}_reverse_a"], 1e-5)
        close(d.ddim_reverse_sample(mdl, xt, t)["sample"], g[f"{name}_reverse_b"], 1e-4)
        ref = g[f"{name}_vb_output"]
        close(d.vb_terms_bpd(mdl, x0, xt, t, clip_denoised=False)["output"] / np.abs(ref).max(), ref / np.abs(ref).max(), 2e-6)
    d = O.create_diffusion_oracle("", learn_sigma=False)
    qm, qv, ql = d.q_posterior(x0, xt, t)
    close(qm, g["qpost_mean"], 1e-6)
    close(qv + torch.zeros_like(xt), g["qpost_var"], 1e-7)
    close(ql + torch.zeros_like(xt), g["qpost_logvar"], 1e-5)
    close(d.predict_xstart_from_eps(xt, t, noise), g["xstart_from_eps"], 1e-3)  # |values| up to ~2e4 at t = 999
    rel = lambda a, b: close(T(a) / np.abs(b).max(), b / np.abs(b).max(), 3e-6)
    d = O.create_diffusion_oracle("", learn_sigma=True)
    for name, lt in (("learned_mse", "mse"), ("learned_rescaled", "rescaled_mse"), ("learned_kl", "rescaled_kl")):
        terms = d.training_losses(_toy2, x0, t, noise, loss_type=lt)
        for k in ("loss", "mse", "vb"):
            if f"{name}_{k}" in g:
                rel(terms[k], g[f"{name}_{k}"])
    d50 = O.create_diffusion_oracle("ddim50", learn_sigma=True)
    terms = d50.training_losses(_toy2, x0, T(g["t50"]), noise)
    rel(terms["loss"], g["ddim50_learned_loss"])
    rel(terms["vb"], g["ddim50_learned_vb"])


def test_eps_tiny(golden):
    g = golden("eps_tiny")
    cfg = TINY_EPS
    sd = O.make_eps_state_dict(cfg, "tiny")
    x, t, lens = T(g["x"]), T(g["t"]), T(g["lens"])
    mask = O.lengths_to_mask(lens, x.shape[1])
    close(O.time_cond(sd, t), g["time_cond"], 1e-5)
    close(O.positional_embedding(mask, cfg.dim), g["pos_emb"], 1e-6)
    tc = O.time_cond(sd, t)
    h = O.causal_conv1d(x, sd["init_conv.weight"], sd["init_conv.bias"])
    close(O.wavenet(O.sub(sd, "wavenet."), h, cfg.wavenet_stacks, cfg.wavenet_layers, tc), g["wavenet"], 2e-5)
    close(O.eps_forward(sd, cfg, x, t, mask), g["eps"], 5e-5)


def test_eps_conditional_variant(golden):
    """use_cond=True: the oracle's PerceiverResampler / prompt conditioning / cross-attention / guidance restatement."""
    from gen_golden_configs import TINY_EPS_COND as cfg

    g = golden("eps_cond_tiny")
    sd = O.make_eps_state_dict(cfg, "cond")
    x, t, lens, plens, prompt = (T(g[k]) for k in ("x", "t", "lens", "plens", "prompt"))
    mask, pmask = O.lengths_to_mask(lens, x.shape[1]), O.lengths_to_mask(plens, prompt.shape[1])
    B = x.shape[0]
    close(O.eps_forward_cond(sd, cfg, x, t, mask, prompt, pmask, torch.zeros(B, dtype=torch.bool)), g["eps_cond"], 2e-5)
    close(O.eps_forward_cond(sd, cfg, x, t, mask, prompt, pmask, torch.ones(B, dtype=torch.bool)), g["eps_null"], 2e-5)
    close(O.eps_forward_with_cond_scale(sd, cfg, x, t, mask, prompt, pmask, 2.0), g["eps_cfg2"], 5e-5)
    masked = prompt.masked_fill(~pmask.unsqueeze(2), 0.0)
    close(O.perceiver_resampler(O.sub(sd, "perceiver_resampler."), masked, pmask, cfg.heads), g["resampled"], 2e-5)


def test_chain_small(golden):
    g = golden("chain_small")
    esd = O.make_eps_state_dict(CHAIN_EPS, "chain")
    vsd = O.make_vae_state_dict(CHAIN_VAE, "chain")
    B, Tn = 3, 48
    feat = seeded((B, Tn, CHAIN_VAE.dim), 31)
    lens, units = T(g["lens"]), T(g["units"])
    mask = O.lengths_to_mask(lens, Tn)
    for start in (1, 5, 50):
        toks, match, total, recon = O.ddim_sample(
            esd, CHAIN_EPS, vsd, CHAIN_VAE, 200, feat, mask, units - 4, start,
            T(g[f"s{start}_post_noise"]), T(g[f"s{start}_start_noise"]))
        close(recon, g[f"s{start}_recon"], 5e-4)
        assert total == int(g[f"s{start}_total"])
        got = torch.cat(toks).numpy()
        agree = (got == g[f"s{start}_units"]).mean()
        assert agree >= 0.99, agree
    ld = O.diffusion_train_forward(esd, CHAIN_EPS, vsd, CHAIN_VAE, 200, feat, units, mask, T(g["train_times"]),
                                   T(g["train_post"]), T(g["train_jitter"]), T(g["train_true"]))
    for k, tol in (("total_loss", 2e-5), ("nll_loss", 2e-4), ("recon_mse_loss", 2e-5), ("noise_loss", 2e-5), ("acc", 0.02)):
        close(ld[k], g["train_" + k], tol)
    mse, logits, kl = O.vae_forward(vsd, CHAIN_VAE, feat, mask, T(g["vae_post"]))
    close(mse, g["vae_mse"], 1e-5)
    close(kl, g["vae_kl"], 1e-5)
    close(logits[:, :8], g["vae_logits_head"], 2e-4)


@pytest.mark.timeout(900)
def test_eps_full_cfg2(golden):
    """BASELINE config 2 shape: full-size eps-predictor [8,256,128], t=500 (about 10 s of CPU)."""
    g = golden("eps_full_cfg2")
    sd = O.make_eps_state_dict(FULL_EPS, "full")
    x = seeded((8, 256, 128), 0)
    lens = T(g["lens"])
    assert lens.tolist() == ragged_lengths(8, 256, 1, lo=128).tolist()
    mask = O.lengths_to_mask(lens, 256)
    close(O.time_cond(sd, T(g["t"])[:1]), g["time_cond"], 1e-4)
    with torch.no_grad():
        eps = O.eps_forward(sd, FULL_EPS, x, T(g["t"]), mask)
    close(eps, g["eps"], 2e-4)


@pytest.mark.timeout(900)
def test_vae_full_cfg1(golden):
    """BASELINE config 1: 64 x [128,768] encode->decode->logits (first 8 utterances re-run here)."""
    g = golden("vae_full_cfg1")
    sd = O.make_vae_state_dict(FULL_VAE, "full")
    n = 8
    feat = seeded((64, 128, 768), 0)[:n]
    lens = T(g["lens"])[:n]
    mask = O.lengths_to_mask(lens, 128)
    noise = seeded((64, 128, 128), 3)[:n]
    with torch.no_grad():
        params = O.vae_encode_params(sd, FULL_VAE, feat)
        z = O.posterior_sample(params, noise)
        recon, logits = O.vae_decode(sd, FULL_VAE, z, mask)
    close(params[:2], g["params_head"], 1e-4)
    close(params.sum(dim=(1, 2)), g["params_sum"][:n], 5e-2)
    close(recon[:2, :, :96], g["recon_head"], 5e-4)
    close(logits[:2, :16], g["logits_head"], 5e-4)
    units = (logits.argmax(-1) - 4).numpy()
    safe = g["margin"][:n] > 1e-3  # argmax is only pinned where the reference's top-2 margin is clear
    assert (units[safe] == g["units"][:n][safe]).all()


def test_oracle_ddpm_chain_matches_the_reference_steps(golden):
    """O.ddpm_chain (the CPU restatement of what dn_ddpm_loop runs) against five real-reference p_sample steps over the real
    reference eps-predictor (tests/golden/ddpm_chain.npz, oracle/gen_golden_ddpm.py): both fixed variances, clip on / off."""
    g = golden("ddpm_chain")
    sd = O.make_eps_state_dict(CHAIN_EPS, "chain")
    lens = torch.from_numpy(g["lens"])
    mask = O.lengths_to_mask(lens, 48)
    for name, var, clip in (("small", "fixed_small", False), ("large", "fixed_large", False), ("small_clip", "fixed_small", True)):
        with torch.no_grad():
            got = O.ddpm_chain(sd, CHAIN_EPS, 200, torch.from_numpy(g[f"{name}_x_start"]), mask, 5, torch.from_numpy(g[f"{name}_noise"]), var, clip)
        err = (got - torch.from_numpy(g[f"{name}_x_end"]))[mask].abs().max().item()
        assert err < 5e-5, (name, err)

