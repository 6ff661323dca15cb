"""GPU parity of the host-side mirror (reference class names / signatures / state-dict keys) and of the fairseq plugin
surface, against the golden vectors of the real reference and the CPU oracle."""
import argparse
import types

import numpy as np
import pytest
import torch

import diffnorm_oracle as O
from gen_golden_configs import CHAIN_EPS, CHAIN_VAE, seeded

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T_(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, tol):
    err = (torch.as_tensor(a).double().cpu() - torch.as_tensor(b).double().cpu()).abs().max().item()
    assert err <= tol, f"max abs err {err} > {tol}"


# every mirror test runs in the three arithmetic modes: exact fp32 and split-operand bf16x3 at the fp32 budget (1e-3, exact
# unit sequences), plain bf16 at the distances the engine-level tests state (tests/test_hip_engine.py)
@pytest.fixture(scope="module", params=["f32", "bf16x3", "bf16"])
def ldm(request):
    from diffnorm_amd.latent_module import LatentDiscreteModel, SpeechVAEEncoderDecoder

    dtype = request.param
    vae = SpeechVAEEncoderDecoder(dim=CHAIN_VAE.dim, latent_dim=CHAIN_VAE.latent_dim, dtype=dtype)
    vsd = O.make_vae_state_dict(CHAIN_VAE, "chain")
    assert set(vae.state_dict()) == set(vsd), "VAE state-dict keys differ from the reference layout"
    vae.load_state_dict(vsd, strict=True)
    m = LatentDiscreteModel(types.SimpleNamespace(encoder=vae), CHAIN_EPS.dim, CHAIN_VAE.z, timesteps=200, dtype=dtype)
    esd = O.make_eps_state_dict(CHAIN_EPS, "chain")
    assert set(m.model.state_dict()) == set(esd) | {"pos_embed._float_tensor"}
    m.model.load_state_dict(dict(esd, **{"pos_embed._float_tensor": torch.zeros(1)}), strict=True)
    # the diffusion checkpoint layout: model.* + speech_decoder.* 
    assert any(k.startswith("speech_decoder.decoder_lm") for k in m.state_dict()) and any(k.startswith("model.wavenet") for k in m.state_dict())
    m.test_dtype = dtype
    return m.to(DEV).eval()


def test_ddim_sample_matches_reference(ldm, golden):
    g = golden("chain_small")
    feat = seeded((3, 48, CHAIN_VAE.dim), 31)
    lens, units = T_(g["lens"]), T_(g["units"])
    mask = O.lengths_to_mask(lens, 48)
    for start in (5, 50):
        toks, match, total, recon = ldm.ddim_sample(feat.to(DEV), input_mask=mask.to(DEV), ref_units=(units - 4).to(DEV),
                                                    start_step=start, post_noise=T_(g[f"s{start}_post_noise"]),
                                                    start_noise=T_(g[f"s{start}_start_noise"]))
        assert total == int(g[f"s{start}_total"])
        if ldm.test_dtype == "bf16":  # measured 1.06-1.23e-2 on these chains (tests/test_hip_engine.py); flat random-init logits flip a few units
            assert (torch.cat(toks).cpu().numpy() == g[f"s{start}_units"]).mean() >= 0.9
            close(recon.cpu()[mask], T_(g[f"s{start}_recon"])[mask], 1.45e-2)
        else:
            assert match == int(g[f"s{start}_match"])
            assert torch.cat(toks).cpu().tolist() == g[f"s{start}_units"].tolist()
            close(recon.cpu()[mask], T_(g[f"s{start}_recon"])[mask], 1e-3)
        assert [t.shape[0] for t in toks] == lens.tolist()


def test_ddpm_sample_through_the_mirror(ldm, golden):
    """LatentDiscreteModel.ddpm_sample (encode -> q_sample to index start_step-1 -> device ancestral loop -> decode): with the
    step noise injected it equals the CPU oracle's chain (encode / p_sample steps / decode, every piece pinned to the reference
    separately); with in-kernel noise it is reproducible per seed."""
    g = golden("chain_small")
    feat = seeded((3, 48, CHAIN_VAE.dim), 31)
    lens = T_(g["lens"])
    mask = O.lengths_to_mask(lens, 48)
    start = 5
    post, sn = T_(g["s5_post_noise"]), T_(g["s5_start_noise"])
    steps = seeded((start, 3, 48, CHAIN_VAE.z), 91)
    toks, _, total, recon = ldm.ddpm_sample(feat.to(DEV), input_mask=mask.to(DEV), start_step=start, post_noise=post, start_noise=sn, step_noise=steps)
    vsd, esd = O.make_vae_state_dict(CHAIN_VAE, "chain"), O.make_eps_state_dict(CHAIN_EPS, "chain")
    with torch.no_grad():
        z = O.vae_encode(vsd, CHAIN_VAE, feat, post)
        tab = O.ddpm_tables(200)
        t = torch.full((3,), start - 1, dtype=torch.long)
        x = tab.at("sqrt_alphas_cumprod", t, 3) * z + tab.at("sqrt_one_minus_alphas_cumprod", t, 3) * sn
        x = O.ddpm_chain(esd, CHAIN_EPS, 200, x, mask, start, steps)
        want_recon, want_logits = O.vae_decode(vsd, CHAIN_VAE, x, mask)
    tol = 1e-3 if ldm.test_dtype != "bf16" else 2e-2
    close(recon.cpu()[mask], want_recon[mask], tol)
    assert total == int(mask.sum()) and [t_.shape[0] for t_ in toks] == lens.tolist()
    if ldm.test_dtype != "bf16":
        want_units = torch.cat([(want_logits[i, : int(lens[i])].argmax(-1) - 4) for i in range(3)])
        assert (torch.cat(toks).cpu() == want_units).float().mean().item() > 0.99
    a = ldm.ddpm_sample(feat.to(DEV), input_mask=mask.to(DEV), start_step=start, post_noise=post, start_noise=sn, seed=3)[3]
    b = ldm.ddpm_sample(feat.to(DEV), input_mask=mask.to(DEV), start_step=start, post_noise=post, start_noise=sn, seed=3)[3]
    c = ldm.ddpm_sample(feat.to(DEV), input_mask=mask.to(DEV), start_step=start, post_noise=post, start_noise=sn, seed=4)[3]
    assert torch.equal(a, b) and not torch.equal(a, c)


def test_reference_rng_draw_order(ldm, golden):
    """Without injected noise the mirror draws the posterior noise from the CPU generator as [B,z,T], like upstream."""
    g = golden("chain_small")
    feat = seeded((3, 48, CHAIN_VAE.dim), 31)
    mask = O.lengths_to_mask(T_(g["lens"]), 48)
    torch.manual_seed(105)  # the generator state the golden run used for start_step=5
    z = ldm.speech_decoder.encode_feature(feat.to(DEV)).transpose(1, 2)
    want = O.vae_encode(O.make_vae_state_dict(CHAIN_VAE, "chain"), CHAIN_VAE, feat, T_(g["s5_post_noise"]))
    close(z, want, 1e-3 if ldm.test_dtype != "bf16" else 1e-2)


def test_training_forward_losses_match_reference(ldm, golden):
    g = golden("chain_small")
    feat = seeded((3, 48, CHAIN_VAE.dim), 31)
    lens, units = T_(g["lens"]), T_(g["units"])
    mask = O.lengths_to_mask(lens, 48)
    with torch.no_grad():
        ld = ldm(feat.to(DEV), units.to(DEV), tgt_mask=mask.to(DEV), times=T_(g["train_times"]), post_noise=T_(g["train_post"]),
                 jitter_noise=T_(g["train_jitter"]), true_noise=T_(g["train_true"]))
    f = 1.0 if ldm.test_dtype != "bf16" else 20.0  # bf16: scalar losses within 2e-2, logits within the engine tests' 2e-2
    for k, tol in (("total_loss", 1e-3 * f), ("nll_loss", 1e-3 * f), ("recon_mse_loss", 1e-3 * f), ("noise_loss", 1e-3 * f), ("acc", 0.02)):
        close(ld[k], g["train_" + k], tol)
    mse, logits, kl = ldm.speech_decoder(feat.to(DEV), units, mask.to(DEV), noise=T_(g["vae_post"]))
    close(mse, g["vae_mse"], 1e-3 * f)
    close(kl, g["vae_kl"], 1e-4 * f)
    close(logits[:, :8], g["vae_logits_head"], 1e-3 * f)


def test_plugin_vae_criterion_end_to_end():
    """--task speech_decoder --arch speech_vae_decoder --criterion speech_vae_decoder_loss on a synthetic batch."""
    from diffnorm_amd.fairseq_plugin import registry as R
    import diffnorm_amd.fairseq_plugin  # noqa: F401

    tp = argparse.ArgumentParser()
    R.TASK_REGISTRY["speech_decoder"].add_args(tp)
    task = R.TASK_REGISTRY["speech_decoder"].setup_task(tp.parse_args(["/data", "--target-code-size", "1000"]))
    mp = argparse.ArgumentParser()
    R.MODEL_REGISTRY["speech_vae_decoder"].add_args(mp)
    margs = mp.parse_args(["--latent_dim", "128", "--hip-dtype", "f32"])
    margs.arch, margs.criterion = "speech_vae_decoder", "speech_vae_decoder_loss"
    model = task.build_model(margs).to(DEV)
    crit = task.build_criterion(margs)
    ds = task.load_dataset("valid", n=4, min_len=20, max_len=40)
    batch = ds.collater([ds[i] for i in range(4)])
    torch.manual_seed(7)
    loss, sample_size, log = task.valid_step(batch, model, crit)
    assert sample_size == 4 and set(log) >= {"loss", "nll_loss", "mse_loss", "kl_loss", "acc", "ntokens", "nsentences"}
    sd = {k[len("encoder."):]: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.manual_seed(7)
    B, T = batch["reduce_target"].shape[:2]
    noise = torch.randn(B, 128, T).transpose(1, 2)
    want = O.vae_criterion(sd, O.VaeConfig(), batch["reduce_target"], batch["reduce_target_unit"], batch["reduce_target_lengths"], noise)
    close(loss, want["loss"], 2e-3)
    close(log["mse_loss"], want["mse_loss"], 1e-3)
    close(log["kl_loss"], want["kl_loss"], 1e-3)
    # task.train_step (criterion forward + optimizer.backward through the HIP training engine): tests/test_hip_train.py


def test_gaussian_diffusion_matches_reference_golden(golden):
    """create_diffusion / q_sample / p_sample / ddim_sample / p_sample_loop / training_losses on the GPU kernels
    against the reference's own outputs (tests/golden/gaussian_diffusion.npz)."""
    from diffnorm_amd.diffusion import create_diffusion

    g = golden("gaussian_diffusion")
    x0, noise, noise2 = (seeded((3, 4, 6), s).to(DEV) for s in (11, 12, 13))
    t = torch.tensor([0, 417, 999], device=DEV)
    pn = T_(g["p_sample_noise"]).to(DEV)
    toy = lambda x, ts, **kw: 0.3 * x - 0.01 * ts.float().view(-1, 1, 1) / 100 + 0.05
    toy2 = lambda x, ts, **kw: torch.cat([toy(x, ts), torch.tanh(x)], dim=1)
    for name, kw, mdl in (("large", dict(learn_sigma=False), toy), ("small", dict(learn_sigma=False, sigma_small=True), toy),
                          ("learned", dict(learn_sigma=True), toy2)):
        d = create_diffusion("", **kw)
        close(d.q_sample(x0, t, noise), g[f"{name}_q_sample"], 1e-6)
        ps = d.p_sample(mdl, x0, t, noise=pn)
        close(ps["sample"], g[f"{name}_p_sample"], 5e-6)
        close(ps["pred_xstart"], g[f"{name}_pred_xstart"], 5e-6)
        close(d.p_sample(mdl, x0, t, clip_denoised=False, noise=pn)["sample"], g[f"{name}_p_sample_noclip"], 5e-5)
        close(d.ddim_sample(mdl, x0, t, noise=pn)["sample"], g[f"{name}_ddim_eta0"], 5e-6)
        close(d.ddim_sample(mdl, x0, t, eta=0.5, noise=pn)["sample"], g[f"{name}_ddim_eta05"], 5e-6)
    d50 = create_diffusion("ddim50", learn_sigma=False)
    assert d50.timestep_map == golden("schedules")["ddim50_timestep_map"].tolist()
    t50 = torch.tensor([0, 20, 49], device=DEV)
    close(d50.p_sample(toy, x0, t50, noise=pn)["sample"], g["ddim50_p_sample"], 5e-6)
    close(d50.q_sample(x0, t50, noise), g["ddim50_q_sample"], 1e-6)
    close(create_diffusion("", learn_sigma=False).training_losses(lambda x, ts, **kw: (toy(x, ts), None), x0, t, noise=noise2)["mse"],
          g["train_mse"], 1e-6)
    # 5-step respaced ancestral loop with the reference's recorded per-step noise
    d5 = create_diffusion("5", learn_sigma=False)
    noises = T_(g["loop5_noises"]).to(DEV)
    x = T_(g["loop5_xT"]).to(DEV)
    for k, i in enumerate(range(4, -1, -1)):
        x = d5.p_sample(toy, x, torch.full((2,), i, device=DEV), noise=noises[k])["sample"]
    close(x, g["loop5_out"], 2e-5)
    # the loop entry point itself (own noise): shape/finite only
    out = d5.p_sample_loop(toy, (2, 4, 6), device=DEV)
    assert out.shape == (2, 4, 6) and torch.isfinite(out).all()


def test_gaussian_diffusion_tail_matches_reference_golden(golden):
    """What is left of GaussianDiffusion behind create_diffusion, against the REAL reference's outputs (tests/golden/
    gaussian_tail.npz, oracle/gen_golden_gd_tail.py): the START_X mean type (predict_xstart=True) through p_sample / p_mean_variance /
    training_losses, cond_fn guidance -- condition_mean in p_sample, condition_score in ddim_sample --, ddim_sample_loop with
    eta > 0, _prior_bpd and calc_bpd_loop (100 steps), and a respaced chain whose model AND cond_fn must see the original step
    indices.  The reference's recorded draws are injected."""
    from diffnorm_amd.diffusion import create_diffusion

    g = golden("gaussian_tail")
    x = T_(g["x"]).to(DEV)
    N, C, L = x.shape

    def toy_model(learned):
        def f(xx, t, **kw):
            tt = t.float().view(-1, 1, 1)
            base = torch.tanh(0.7 * xx + 0.01 * tt) - 0.1 * xx
            return torch.cat([base, torch.sin(1.3 * xx + 0.02 * tt)], dim=1) if learned else base
        return f

    cond_fn = lambda xx, t, **kw: 0.3 * torch.cos(xx) - 0.001 * t.float().view(-1, 1, 1)
    rel = lambda a, b, tol: close(torch.as_tensor(a).cpu().double() / max(np.abs(b).max(), 1e-6), b / max(np.abs(b).max(), 1e-6), tol)
    t = torch.tensor([0, 37, 99], device=DEV)
    for name, kw, learned in (("sx", dict(predict_xstart=True, learn_sigma=False), False),
                              ("sx_lr", dict(predict_xstart=True, learn_sigma=True), True),
                              ("eps", dict(learn_sigma=False, sigma_small=True), False)):
        d = create_diffusion("", diffusion_steps=100, **kw)
        model = toy_model(learned)
        for clip in (False, True):
            tag = f"{name}_clip{int(clip)}"
            o = d.p_sample(model, x, t, clip_denoised=clip, noise=T_(g[f"{tag}_p_noise"]).to(DEV))
            rel(o["sample"], g[f"{tag}_p_sample"], 1e-5)
            rel(o["pred_xstart"], g[f"{tag}_p_x0"], 1e-5)
            o = d.p_sample(model, x, t, clip_denoised=clip, cond_fn=cond_fn, model_kwargs={}, noise=T_(g[f"{tag}_pc_noise"]).to(DEV))
            rel(o["sample"], g[f"{tag}_pc_sample"], 1e-5)
            o = d.ddim_sample(model, x, t, clip_denoised=clip, cond_fn=cond_fn, model_kwargs={}, eta=0.5, noise=T_(g[f"{tag}_dc_noise"]).to(DEV))
            rel(o["sample"], g[f"{tag}_dc_sample"], 2e-5)
            rel(o["pred_xstart"], g[f"{tag}_dc_x0"], 2e-5)
        pm = d.p_mean_variance(model, x, t, clip_denoised=True)
        rel(pm["mean"], g[f"{name}_pmv_mean"], 1e-5)
        rel(pm["log_variance"], g[f"{name}_pmv_logvar"], 1e-5)
        # denoised_fn (applies to the x_0 prediction before the clip, :263-265): the un-fused fallback of p_mean_variance / p_sample
        # (with cond_fn) / ddim_sample (eta 0.5) / ddim_reverse_sample and the loop that threads it through
        dfn = lambda z: 0.8 * torch.tanh(1.5 * z)
        for clip in (False, True):
            tag = f"{name}_dfn_clip{int(clip)}"
            pm = d.p_mean_variance(model, x, t, clip_denoised=clip, denoised_fn=dfn)
            rel(pm["mean"], g[f"{tag}_pmv_mean"], 1e-5)
            rel(pm["pred_xstart"], g[f"{tag}_pmv_x0"], 1e-5)
            o = d.p_sample(model, x, t, clip_denoised=clip, denoised_fn=dfn, cond_fn=cond_fn, model_kwargs={}, noise=T_(g[f"{tag}_p_noise"]).to(DEV))
            rel(o["sample"], g[f"{tag}_p_sample"], 1e-5)
            rel(o["pred_xstart"], g[f"{tag}_p_x0"], 1e-5)
            o = d.ddim_sample(model, x, t, clip_denoised=clip, denoised_fn=dfn, eta=0.5, noise=T_(g[f"{tag}_d_noise"]).to(DEV))
            rel(o["sample"], g[f"{tag}_d_sample"], 2e-5)
            rel(o["pred_xstart"], g[f"{tag}_d_x0"], 2e-5)
            o = d.ddim_reverse_sample(model, x, t, clip_denoised=clip, denoised_fn=dfn, eta=0.0)
            rel(o["sample"], g[f"{tag}_r_sample"], 2e-5)
            rel(o["pred_xstart"], g[f"{tag}_r_x0"], 2e-5)
        dn = T_(g[f"{name}_dfn_dloop_noises"]).to(DEV)
        img = dn[0]
        for k, i in enumerate(range(99, -1, -1)):
            img = d.ddim_sample(model, img, torch.tensor([i] * N, device=DEV), clip_denoised=True, denoised_fn=dfn, eta=0.3, noise=dn[k + 1])["sample"]
        rel(img, g[f"{name}_dfn_dloop_out"], 1e-4)
        tl = d.training_losses(lambda *a, **k: (model(*a, **k), None), x, t, noise=T_(g[f"{name}_tl_noise"]).to(DEV))
        for k in ("loss", "mse", "vb"):
            if f"{name}_tl_{k}" in g:
                rel(tl[k], g[f"{name}_tl_{k}"], 3e-5)
        rel(d._prior_bpd(x), g[f"{name}_prior_bpd"], 1e-5)
        noises = T_(g[f"{name}_bpd_noises"])  # drawn for t = 99 .. 0
        bp = d.calc_bpd_loop(model, x, clip_denoised=True, noises=[noises[99 - i] for i in range(100)])
        for k in ("total_bpd", "prior_bpd", "vb", "xstart_mse", "mse"):
            rel(bp[k], g[f"{name}_bpd_{k}"], 5e-4 if k == "mse" else 5e-5)  # (eps re-derived from x_0 divides by sqrt(1/abar - 1) -> tiny at t = 0)
        # ddim_sample_loop with injected per-step noise: the loop is this class's own, the draws come from the fixture
        dn = T_(g[f"{name}_dloop_noises"]).to(DEV)  # the recorder also caught the initial noise (draw 0); draws 1 .. 100 are the steps'
        assert dn.shape[0] == 101 and torch.equal(dn[0].cpu(), seeded((N, C, L), 502))
        img = dn[0]
        for k, i in enumerate(range(99, -1, -1)):
            img = d.ddim_sample(model, img, torch.tensor([i] * N, device=DEV), clip_denoised=True, eta=0.3, noise=dn[k + 1])["sample"]
        rel(img, g[f"{name}_dloop_out"], 1e-4)
    # the loop method itself (eta = 0: no draw matters) on the respaced diffusion with cond_fn
    d = create_diffusion("5", diffusion_steps=100, learn_sigma=False)
    y = d.ddim_sample_loop(toy_model(False), (N, C, L), noise=seeded((N, C, L), 503).to(DEV), clip_denoised=False, cond_fn=cond_fn, model_kwargs={},
                           eta=0.0)
    rel(y, g["resp_dloop_out"], 1e-4)


def test_gaussian_moments_match_reference_golden(golden):
    """The rest of GaussianDiffusion on the GPU kernel dn_gaussian_moments against the reference's outputs
    (tests/golden/gaussian_moments.npz): q_posterior_mean_variance, p_mean_variance x 3 variance types x clip on/off,
    _predict_xstart_from_eps, ddim_reverse_sample, _vb_terms_bpd and training_losses with a learned variance (create_diffusion's
    DEFAULT), rescaled, KL, and respaced."""
    from diffnorm_amd.diffusion import create_diffusion

    g = golden("gaussian_moments")
    x0, xt, noise = (T_(g[k]).to(DEV) for k in ("x0", "xt", "noise"))
    t = T_(g["t"]).to(DEV)
    toy = lambda x, ts, **kw: 0.3 * x - 0.01 * ts.float().view(-1, 1, 1) / 100 + 0.05
    toy2 = lambda x, ts, **kw: torch.cat([toy(x, ts), torch.tanh(x)], dim=1)
    rel = lambda a, b, tol: close(a.cpu().double() / np.abs(b).max(), b / np.abs(b).max(), tol)
    for name, kw, mdl in (("large", dict(learn_sigma=False), toy), ("small", dict(learn_sigma=False, sigma_small=True), toy),
                          ("learned", dict(learn_sigma=True), toy2)):
        d = create_diffusion("", **kw)
        for clip in (True, False):
            pm = d.p_mean_variance(mdl, xt, t, clip_denoised=clip)
            for k in ("mean", "variance", "log_variance", "pred_xstart"):
                rel(pm[k], g[f"{name}_pmv{int(clip)}_{k}"], 5e-6)
        rel(d.ddim_reverse_sample(mdl, xt, T_(g["t_reverse_a"]).to(DEV))["sample"], g[f"{name}_reverse_a"], 5e-6)
        rel(d.ddim_reverse_sample(mdl, xt, t)["sample"], g[f"{name}_reverse_b"], 5e-6)
        rel(d._vb_terms_bpd(mdl, x0, xt, t, clip_denoised=False)["output"], g[f"{name}_vb_output"], 2e-5)
    d = create_diffusion("", learn_sigma=False)
    qm, qv, ql = d.q_posterior_mean_variance(x0, xt, t)
    rel(qm, g["qpost_mean"], 2e-6)
    rel(qv, g["qpost_var"], 2e-6)
    rel(ql, g["qpost_logvar"], 2e-6)
    rel(d._predict_xstart_from_eps(xt, t, noise), g["xstart_from_eps"], 2e-6)
    tl = lambda x, ts, **kw: (toy2(x, ts), None)
    for name, kw, mdl in (("learned_mse", dict(learn_sigma=True), tl), ("learned_rescaled", dict(learn_sigma=True, rescale_learned_sigmas=True), tl),
                          ("learned_kl", dict(learn_sigma=True, use_kl=True), toy2)):
        terms = create_diffusion("", **kw).training_losses(mdl, x0, t, noise=noise)  # learn_sigma=True is create_diffusion's default
        for k in ("loss", "mse", "vb"):
            if f"{name}_{k}" in g:
                rel(terms[k], g[f"{name}_{k}"], 2e-5)
    terms = create_diffusion("ddim50").training_losses(tl, x0, T_(g["t50"]).to(DEV), noise=noise)
    rel(terms["loss"], g["ddim50_learned_loss"], 2e-5)
    rel(terms["vb"], g["ddim50_learned_vb"], 2e-5)


def test_conditional_variant_through_the_mirror(golden):
    """use_cond=True through the reference-named modules: Model(condition_on_prompt=True) with the reference's
    state-dict keys, forward / forward_with_cond_scale against the reference's outputs, and a prompted, guided DDIM chain of
    LatentDiscreteModel against the oracle's per-step restatement."""
    from diffnorm_amd import ops, scheduler
    from diffnorm_amd.latent_module import LatentDiscreteModel, Model, SpeechVAEEncoderDecoder
    from gen_golden_configs import TINY_EPS_COND as cfg

    g = golden("eps_cond_tiny")
    sd = O.make_eps_state_dict(cfg, "cond")
    m = Model(cfg.dim, cfg.latent_dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads, wavenet_layers=cfg.wavenet_layers,
              wavenet_stacks=cfg.wavenet_stacks, condition_on_prompt=True, dim_prompt=cfg.dim_prompt, num_latents_m=cfg.num_latents_m,
              resampler_depth=cfg.resampler_depth, dtype="f32")
    extra = {"pos_embed._float_tensor", "perceiver_resampler.embed_positions._float_tensor"}
    assert set(m.state_dict()) == set(sd) | extra  # the reference's conditional key set (strict-loaded there by gen_golden.py)
    m.load_state_dict(dict(sd, **{k: torch.zeros(1) for k in extra}), strict=True)
    m.to(DEV)
    x, t, lens, plens, prompt = (T_(g[k]) for k in ("x", "t", "lens", "plens", "prompt"))
    mask, pmask = O.lengths_to_mask(lens, 40), O.lengths_to_mask(plens, 21)
    kw = dict(prompt=prompt.to(DEV), prompt_mask=pmask.to(DEV), input_mask=mask.to(DEV))
    close(m(x.to(DEV), t, cond_drop_prob=0.0, **kw).cpu()[mask], T_(g["eps_cond"])[mask], 1e-4)
    close(m(x.to(DEV), t, cond_drop_prob=1.0, **kw).cpu()[mask], T_(g["eps_null"])[mask], 1e-4)
    close(m.forward_with_cond_scale(x.to(DEV), t, cond_scale=2.0, **kw).cpu()[mask], T_(g["eps_cfg2"])[mask], 1e-4)
    with pytest.raises(ValueError):
        m(x.to(DEV), t, input_mask=mask.to(DEV))
    # prompted + guided chain: LatentDiscreteModel(use_cond=True) builds Model(dim, z, condition_on_prompt=True, dim_prompt = feature dim)
    vae = SpeechVAEEncoderDecoder(dim=CHAIN_VAE.dim, latent_dim=CHAIN_VAE.latent_dim, dtype="f32")
    vsd = O.make_vae_state_dict(CHAIN_VAE, "chain")
    vae.load_state_dict(vsd, strict=True)
    ldm = LatentDiscreteModel(types.SimpleNamespace(encoder=vae), 64, CHAIN_VAE.z, timesteps=200, use_cond=True, dtype="f32").to(DEV).eval()
    assert ldm.model.condition_on_prompt and ldm.model.cfg.dim_prompt == CHAIN_VAE.dim and ldm.model.cfg.num_latents_m == 64
    ecfg = O.EpsConfig(dim=64, latent_dim=CHAIN_VAE.z, dim_prompt=CHAIN_VAE.dim, num_latents_m=64)
    esd = {k: v.detach().cpu() for k, v in ldm.model.state_dict().items() if not k.endswith("._float_tensor")}
    feat, src = seeded((2, 24, CHAIN_VAE.dim), 91), seeded((2, 30, CHAIN_VAE.dim), 92)
    flen, slen = torch.tensor([24, 15]), torch.tensor([30, 22])
    fmask, smask = O.lengths_to_mask(flen, 24), O.lengths_to_mask(slen, 30)
    post, start = seeded((2, 24, CHAIN_VAE.z), 93), seeded((2, 24, CHAIN_VAE.z), 94)
    toks, _, total, recon = ldm.ddim_sample(feat.to(DEV), prompt=src.to(DEV), prompt_mask=smask.to(DEV), input_mask=fmask.to(DEV),
                                            cond_scale=2.0, start_step=4, post_noise=post, start_noise=start)
    tab = O.ddpm_tables(200)
    xx = O.vae_encode(vsd, CHAIN_VAE, feat, post)
    ts = torch.full((2,), 4, dtype=torch.long)
    xx = tab.at("sqrt_alphas_cumprod", ts, 3) * xx + tab.at("sqrt_one_minus_alphas_cumprod", ts, 3) * start
    for time in (3, 2, 1):
        tt = torch.full((2,), time, dtype=torch.long)
        xx = O.ddim_update(tab, xx, O.eps_forward_with_cond_scale(esd, ecfg, xx, tt, fmask, src, smask, 2.0), tt)
    want, _ = O.vae_decode(vsd, CHAIN_VAE, xx, fmask)
    close(recon.cpu()[fmask], want[fmask], 2e-3)
    assert total == int(flen.sum()) and [t_.shape[0] for t_ in toks] == flen.tolist()
