"""GPU checks at BASELINE.json's full sizes through size-independent properties (configs 3 and 5) plus a truncated
chain against the CPU oracle on a slice of the batch."""
import numpy as np
import pytest
import torch

import diffnorm_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def models():
    from diffnorm_amd import engine, scheduler, synthetic

    cfg = synthetic.eps_config()
    esd = synthetic.random_eps_state_dict(cfg, seed=0)
    vsd = synthetic.random_vae_state_dict(seed=1)
    return dict(cfg=cfg, esd=esd, vsd=vsd, eps=engine.EpsEngine(esd, cfg, dtype="bf16", device=DEV),
                vae=engine.VaeEngine(vsd, dtype="bf16", device=DEV), sched=scheduler.DDPMScheduler(1000))


def test_config3_chain_graph_and_split_are_bit_identical(models):
    """[B=32,T=512] latents, 1000-step schedule: eager == hipGraph replay == forked half-batch replay, bit for bit,
    and the chain stays finite (a 12-step slice of the full chain, t = 998..987)."""
    from diffnorm_amd import ops

    eps, sched = models["eps"], models["sched"]
    coef = sched.ddim_coef_table(DEV)
    x0 = ops.randn((32, 512, 128), seed=5, device=DEV)
    lens = torch.full((32,), 512, dtype=torch.int32, device=DEV)
    lens[3], lens[17] = 300, 1
    outs = []
    for graph, split in ((False, False), (True, False), (True, True), (False, True)):
        x = x0.clone()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            n = eps.ddim_loop(x, lens, 999, coef, use_graph=graph, max_evals=12, split=split)
        torch.cuda.synchronize()
        assert n == 12 and torch.isfinite(x).all()
        outs.append(x)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    assert not torch.equal(outs[0], x0)


def test_config5_end_to_end_sharding_invariance_and_oracle_slice(models, hip_option):
    """[B=16,T=1024] features: VAE encode -> noise at start_step -> DDIM -> VAE decode -> units.  (a) the 16-utterance
    batch and two 8-utterance shards (the multi-GPU sharding): with every contraction in term-outer K order
    (DN_TAPS_INNER=0) units and recon are identical bit for bit whatever tile variants the shard sizes route to; in the
    default order the 256x352 tile sums the FFN conv's taps innermost, so shards small enough to leave that tile (here the
    half-batch streams of an 8-utterance shard: M = 4096 rows) differ from the big batch in the last bits of fp32 sums --
    units agree but for rare near-ties (<= 1 %), recon within the bf16 rounding level (3e-2 of its scale); with the tap contractions
    routed by shape (DN_TAPS_INNER=2) the fast order is invariant too; (b) a truncated chain (start_step=3) on 2 utterances matches the CPU oracle
    within the bf16 budget and agrees on units where the oracle's top-2 margin is clear."""
    from diffnorm_amd import ops

    eps, vae, sched, cfg = models["eps"], models["vae"], models["sched"], models["cfg"]
    coef = sched.ddim_coef_table(DEV)
    sa, s1 = sched.f32("sqrt_alphas_cumprod", DEV), sched.f32("sqrt_one_minus_alphas_cumprod", DEV)
    B, T, start = 16, 1024, 3
    g = torch.Generator().manual_seed(0)
    feat = torch.randn(B, T, 768, generator=g)
    lens = torch.randint(400, T + 1, (B,), generator=g)
    lens[0] = T
    post = torch.randn(B, T, 128, generator=g)
    noise = torch.randn(B, T, 128, generator=g)

    def run(sl):
        f, l = feat[sl].to(DEV), lens[sl]
        z = vae.sample_posterior(vae.encode_params(f), post[sl])
        ts = torch.full((f.shape[0],), start, dtype=torch.int32, device=DEV)
        x = ops.q_sample(z, noise[sl].to(DEV), sa, s1, ts, T)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            eps.ddim_loop(x, l.to(DEV).int(), start, coef, use_graph=False)
        torch.cuda.synchronize()
        recon, logits, units = vae.decode(x, l)
        return recon.cpu(), logits.cpu(), units.cpu()

    hip_option("taps_inner", 0)
    full = run(slice(0, 16))
    a, b = run(slice(0, 8)), run(slice(8, 16))
    assert torch.equal(torch.cat([a[2], b[2]]), full[2])
    assert torch.equal(torch.cat([a[0], b[0]]), full[0])
    hip_option("taps_inner", None)
    strict = full
    full = run(slice(0, 16))
    # DN_TAPS_INNER=2 (the sharded driver's default): tap contractions routed by shape -> the fast order AND bit-for-bit invariance,
    # down to single-utterance shards (the big batch's narrow convs leave the 128-byte-K-tile kernels too, so it need not equal `full`)
    hip_option("taps_inner", 2)
    routed = run(slice(0, 16))
    parts = [run(slice(0, 8)), run(slice(8, 15)), run(slice(15, 16))]
    assert torch.equal(torch.cat([q[2] for q in parts]), routed[2])
    assert torch.equal(torch.cat([q[0] for q in parts]), routed[0])
    hip_option("taps_inner", None)
    a, b = run(slice(0, 8)), run(slice(8, 16))
    valid = torch.arange(T)[None, :] < lens[:, None]
    for got in (torch.cat([a[2], b[2]]), strict[2]):  # shards vs batch, and the batch in the other K order
        flips = (got != full[2])[valid].float().mean().item()
        print(f"config5: units that differ between K orders / shardings: {flips:.2e}")
        assert flips <= 1e-2
    scale = full[0][valid].abs().max().item()
    for got in (torch.cat([a[0], b[0]]), strict[0]):  # last-bit fp32 differences flip a few bf16 roundings downstream
        d = (got - full[0])[valid].abs().max().item()
        print(f"config5: recon max abs difference {d:.3e} of scale {scale:.3e}")
        assert d <= 3e-2 * scale
    # oracle on utterances 0..1 (full length 1024 and a ragged one)
    n = 2
    ocfg = O.EpsConfig()
    mask = O.lengths_to_mask(lens[:n], T)
    with torch.no_grad():
        units_o, _, _, recon_o, = O.ddim_sample(models["esd"], ocfg, models["vsd"], O.VaeConfig(), 1000, feat[:n], mask,
                                                torch.zeros(n, T, dtype=torch.long), start, post[:n], noise[:n])
        # margins of the oracle's logits for a fair unit comparison
    rec = full[0][:n]
    err = (rec - recon_o)[mask].abs().max().item()
    rel = err / recon_o[mask].abs().max().item()
    print(f"config5 slice: recon max abs err {err:.3e} (rel {rel:.3e})")
    assert rel < 5e-2
    got = [full[2][i, : int(lens[i])] for i in range(n)]
    agree = np.mean([(g_.long() == u_).float().mean().item() for g_, u_ in zip(got, units_o)])
    print(f"config5 slice: unit agreement with the oracle {agree:.4f}")
    assert agree > 0.9


def test_bf16x3_fullsize_chain_is_shard_invariant_and_matches_the_oracle(models):
    """The split-operand mode at BASELINE's full sizes.  (a) configs[2] shape [32,512]: eager == hipGraph == forked half-batch
    replay, bit for bit.  (b) configs[4] shape [16,1024] end to end: every split-operand tile sums K in the same (term-outer)
    order, so the 16-utterance batch and its two 8-utterance shards -- the multi-GPU sharding -- agree BIT FOR BIT in the default
    configuration (no environment switch), whatever tiles the shard sizes route to.  (c) the truncated chain on two utterances
    against the CPU oracle at the fp32 budget: reconstruction within 1e-3 of its scale, units identical where the oracle's
    top-2 logit margin is clear."""
    from diffnorm_amd import engine, ops

    sched, cfg = models["sched"], models["cfg"]
    eps = engine.EpsEngine(models["esd"], cfg, dtype="bf16x3", device=DEV)
    vae = engine.VaeEngine(models["vsd"], dtype="bf16x3", device=DEV)
    coef = sched.ddim_coef_table(DEV)
    x0 = ops.randn((32, 512, 128), seed=5, device=DEV)
    lens32 = torch.full((32,), 512, dtype=torch.int32, device=DEV)
    lens32[3], lens32[17] = 300, 1
    outs = []
    for graph, split in ((False, False), (True, False), (True, True)):
        x = x0.clone()
        with torch.cuda.stream(torch.cuda.Stream()):
            assert eps.ddim_loop(x, lens32, 999, coef, use_graph=graph, max_evals=6, split=split) == 6
        torch.cuda.synchronize()
        assert torch.isfinite(x).all()
        outs.append(x)
    assert torch.equal(outs[1], outs[0]) and torch.equal(outs[2], outs[0]) and not torch.equal(outs[0], x0)

    sa, s1 = sched.f32("sqrt_alphas_cumprod", DEV), sched.f32("sqrt_one_minus_alphas_cumprod", DEV)
    B, T, start = 16, 1024, 3
    g = torch.Generator().manual_seed(0)
    feat = torch.randn(B, T, 768, generator=g)
    lens = torch.randint(400, T + 1, (B,), generator=g)
    lens[0] = T
    post = torch.randn(B, T, 128, generator=g)
    noise = torch.randn(B, T, 128, generator=g)

    def run(sl):
        f, l = feat[sl].to(DEV), lens[sl]
        z = vae.sample_posterior(vae.encode_params(f), post[sl])
        ts = torch.full((f.shape[0],), start, dtype=torch.int32, device=DEV)
        x = ops.q_sample(z, noise[sl].to(DEV), sa, s1, ts, T)
        with torch.cuda.stream(torch.cuda.Stream()):
            eps.ddim_loop(x, l.to(DEV).int(), start, coef, use_graph=False)
        torch.cuda.synchronize()
        recon, logits, units = vae.decode(x, l)
        return recon.cpu(), logits.cpu(), units.cpu()

    full = run(slice(0, 16))
    a, b = run(slice(0, 8)), run(slice(8, 16))
    assert torch.equal(torch.cat([a[2], b[2]]), full[2])
    assert torch.equal(torch.cat([a[0], b[0]]), full[0]) and torch.equal(torch.cat([a[1], b[1]]), full[1])
    n = 2
    mask = O.lengths_to_mask(lens[:n], T)
    with torch.no_grad():
        units_o, _, _, recon_o, = O.ddim_sample(models["esd"], O.EpsConfig(), models["vsd"], O.VaeConfig(), 1000, feat[:n], mask,
                                                torch.zeros(n, T, dtype=torch.long), start, post[:n], noise[:n])
    err = (full[0][:n] - recon_o)[mask].abs().max().item()
    scale = recon_o[mask].abs().max().item()
    print(f"bf16x3 config5 slice: recon max abs err {err:.3e} of scale {scale:.3e}")
    assert err < 1e-3 * max(scale, 1.0)
    got = [full[2][i, : int(lens[i])] for i in range(n)]
    agree = np.mean([(g_.long() == u_).float().mean().item() for g_, u_ in zip(got, units_o)])
    print(f"bf16x3 config5 slice: unit agreement with the oracle {agree:.4f}")
    assert agree > 0.995


def test_manifests_to_normalised_unit_tsv_through_the_hip_path(tmp_path):
    """SURVEY 8 f1 on the GPU: on-disk unit TSVs + per-utterance feature .npy files -> data.load_normalization_inputs ->
    normalize() driving the REAL LatentDiscreteModel.ddim_sample on the HIP engines (pinned staging buffer + async H2D per batch,
    two batches) -> normalised-unit TSV, against the CPU oracle's ddim_sample on the same batches with the same noise (f32)."""
    import types

    from diffnorm_amd import data as D
    from diffnorm_amd import normalize as N
    from diffnorm_amd.latent_module import LatentDiscreteModel, SpeechVAEEncoderDecoder
    from gen_golden_configs import CHAIN_EPS, CHAIN_VAE

    rng = np.random.RandomState(3)
    for d in ("orig", "reduce", "feat/dev"):
        (tmp_path / d).mkdir(parents=True)
    rows_o, rows_r, feats, frames = [N.TSV_HEADER], [N.TSV_HEADER], {}, {}
    for i, n_runs in enumerate((21, 34, 27)):
        runs = rng.randint(1, 4, size=n_runs)
        vals = rng.randint(0, 1000, size=n_runs)
        vals[1:][vals[1:] == vals[:-1]] += 1  # neighbouring runs differ, so de-duplication is exactly the run heads
        units = np.repeat(vals, runs)
        uid = f"utt{i}"
        rows_o.append(f"{uid}\tsrc{i}.wav\t{100 + i}\t{' '.join(map(str, units))}\t{len(units)}")
        rows_r.append(f"{uid}\tsrc{i}.wav\t{100 + i}\t{' '.join(map(str, vals))}\t{n_runs}")
        feats[uid] = rng.randn(len(units), CHAIN_VAE.dim).astype(np.float32)
        frames[uid] = np.cumsum(np.concatenate([[0], runs[:-1]]))
        np.save(tmp_path / "feat" / "dev" / f"{uid}.feat.npy", feats[uid])
    (tmp_path / "orig" / "dev.tsv").write_text("\n".join(rows_o) + "\n")
    (tmp_path / "reduce" / "dev.tsv").write_text("\n".join(rows_r) + "\n")
    utts = D.load_normalization_inputs(str(tmp_path / "reduce"), str(tmp_path / "orig"), str(tmp_path / "feat"), "dev")
    assert [u.audio_id for u in utts] == ["utt0", "utt1", "utt2"] and all(isinstance(u.feat, str) for u in utts)

    vsd, esd = O.make_vae_state_dict(CHAIN_VAE, "chain"), O.make_eps_state_dict(CHAIN_EPS, "chain")
    vae = SpeechVAEEncoderDecoder(dim=CHAIN_VAE.dim, latent_dim=CHAIN_VAE.latent_dim, dtype="f32")
    vae.load_state_dict(vsd, strict=True)
    ldm = LatentDiscreteModel(types.SimpleNamespace(encoder=vae), CHAIN_EPS.dim, CHAIN_VAE.z, timesteps=200, dtype="f32")
    ldm.model.load_state_dict(dict(esd, **{"pos_embed._float_tensor": torch.zeros(1)}), strict=True)
    ldm = ldm.to(DEV).eval()
    calls = []

    def sample(feat, input_mask, cond_scale, ref_units, start_step):
        g = torch.Generator().manual_seed(500 + len(calls))
        post = torch.randn(feat.shape[0], feat.shape[1], CHAIN_VAE.z, generator=g)
        start = torch.randn(feat.shape[0], feat.shape[1], CHAIN_VAE.z, generator=g)
        calls.append((feat.cpu(), input_mask.cpu(), ref_units.cpu(), post, start))
        return ldm.ddim_sample(feat, input_mask=input_mask, cond_scale=cond_scale, ref_units=ref_units, start_step=start_step,
                               post_noise=post, start_noise=start)

    lines = N.normalize(sample, utts, start_step=5, batch_size=2, device=DEV)
    out = tmp_path / "dev.normalized.tsv"
    D.write_unit_tsv(str(out), lines)
    assert len(calls) == 2 and calls[0][0].shape[0] == 2 and calls[1][0].shape[0] == 1
    # the assembled batch is what the reference assembles: the first frame of every unit run, zero padded
    assert torch.equal(calls[0][0][0, :21], torch.from_numpy(feats["utt0"][frames["utt0"]]))
    assert calls[0][0][0, 21:].abs().max().item() == 0
    want_lines = []
    agree = tot = 0
    for (feat, mask, ref, post, start), items in zip(calls, (utts[:2], utts[2:])):
        units, _, _, _ = O.ddim_sample(esd, CHAIN_EPS, vsd, CHAIN_VAE, 200, feat, mask, ref, 5, post, start)
        want_lines += [N.tsv_line(it, u.tolist()) for it, u in zip(items, units)]
    got_rows = D.read_unit_tsv(str(out))
    for w in want_lines:
        uid, src, n, un, cnt = w.split("\t")
        g_src, g_n, g_units, g_cnt = got_rows[uid]
        assert (g_src, g_n) == (src, int(n))
        a, b = un.split(" "), g_units.split(" ")
        agree += sum(x == y for x, y in zip(a, b)) if len(a) == len(b) else 0
        tot += len(a)
    assert agree / tot >= 0.99, (agree, tot)  # f32: identical except where the reference's top-2 logit margin is below round-off
    assert open(out).readline().strip() == N.TSV_HEADER


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-3), ("bf16x3", 1e-3), ("f16", 1e-2), ("bf16", 4e-2)])
def test_fullsize_conditional_variant_vs_oracle_and_captured_guided_chain(dtype, tol):
    """SURVEY 8 f3 at the recipe's sizes -- Model(512, 128, condition_on_prompt=True, dim_prompt=768, num_latents_m=64): twice the
    conditioning width (4096), the PerceiverResampler over a ragged 768-wide prompt and a cross-attention block in each of the 12
    layers.  (a) the guided prediction (cond_scale 2: conditioned and null rows in ONE 2B-row pass) against the CPU oracle on a
    [2, 64] batch; (b) a 6-step prompted, guided DDIM chain: one step captured into a hipGraph and replayed equals the eager
    host-stepped chain bit for bit."""
    from diffnorm_amd import engine, scheduler

    cfg = O.EpsConfig(dim_prompt=768, num_latents_m=64)
    sd = O.make_eps_state_dict(cfg, "condfull")
    e = engine.EpsEngine(sd, cfg, dtype=dtype, device=DEV)
    g = torch.Generator().manual_seed(7)
    B, T, Tp = 2, 64, 48
    x = torch.randn(B, T, 128, generator=g)
    prompt = torch.randn(B, Tp, 768, generator=g)
    lens, plens = torch.tensor([64, 37]), torch.tensor([48, 20])
    t = torch.tensor([7, 7])  # (small t: O(1) FiLM / adaptive-norm gains, the regime where plain bf16 keeps its budget)
    mask, pmask = O.lengths_to_mask(lens, T), O.lengths_to_mask(plens, Tp)
    with torch.no_grad():
        want = O.eps_forward_with_cond_scale(sd, cfg, x, t, mask, prompt, pmask, 2.0)
    got = e.forward_with_cond_scale(x.to(DEV), t, lens, prompt.to(DEV), plens, cond_scale=2.0).cpu()
    err = (got - want)[mask].abs().max().item()
    print(f"full-size conditional variant, guided (scale 2) {dtype}: max abs err {err:.3e}")
    assert err < tol
    coef = scheduler.DDPMScheduler(200).ddim_coef_table(DEV)
    outs = []
    for graph in (False, True):
        xs = x.to(DEV).clone()
        with torch.cuda.stream(torch.cuda.Stream()):
            assert e.guided_ddim_chain(xs, lens, prompt.to(DEV), plens, 7, coef, cond_scale=2.0, use_graph=graph) == 6
        torch.cuda.synchronize()
        assert torch.isfinite(xs).all()
        outs.append(xs.cpu())
    assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], x)
