"""GPU parity of the VAE training step (SURVEY 8 f2 / BASELINE config 4) against fixtures taken from the REAL reference
(oracle/gen_golden_train.py -> tests/golden/vae_train.npz: autograd gradients of every parameter + a 5-update trajectory driven
like fairseq's trainer with the reference's own Adam / clip_grad_norm_ / inverse_sqrt classes).  Exact-fp32 mode carries the
north_star budget (1e-3); bf16 mode is held to the oracle's gradients by direction and norm."""
import math

import numpy as np
import pytest
import torch

import diffnorm_oracle as O
import train_oracle as TO
from gen_golden_configs import CHAIN_VAE, seeded

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
CFG = CHAIN_VAE


def _engine(dtype, sd=None):
    from diffnorm_amd import training

    sd = O.make_vae_state_dict(CFG, "train") if sd is None else sd
    return training.VaeTrainEngine(sd, dim=CFG.dim, latent_dim=CFG.latent_dim, dtype=dtype, device=DEV, depth=CFG.depth,
                                   heads=CFG.heads, dim_head=CFG.dim_head, stacks=CFG.stacks, layers=CFG.layers), sd


def _batch(g):
    feat = seeded((3, 48, CFG.dim), 31)
    return feat, torch.from_numpy(g["units"]), torch.from_numpy(g["lens"])


def test_state_dict_round_trip_through_the_packed_layout():
    eng, sd = _engine("f32")
    back = eng.state_dict()
    assert set(back) == set(sd)
    for k in sd:
        assert torch.equal(back[k], sd[k].float()), k
    ranges = eng.stage_ranges()
    assert sum(c for _, c in ranges) == eng.n_params and ranges[-1][0] == 0
    for (o0, c0), (o1, c1) in zip(ranges[1:], ranges[:-1]):
        assert o0 + c0 == o1  # descending, contiguous: the backward walks the buffer from its end


def test_vae_losses_and_gradients_match_reference_f32(golden):
    g = golden("vae_train")
    feat, units, lens = _batch(g)
    eng, _ = _engine("f32")
    stats, logits, _ = eng.forward(feat, units, lens, noise=torch.from_numpy(g["post_noise"]), ntokens=int(lens.sum()), want_logits=True)
    eng.zero_grad()
    eng.backward()
    s = stats.cpu().double().numpy()
    for i, k in enumerate(("loss", "nll_loss", "mse_loss", "kl_loss", "acc")):
        assert abs(s[i] - float(g[k])) <= 1e-4 * max(1.0, abs(float(g[k]))), (k, s[i], float(g[k]))
    assert np.abs(logits.cpu().numpy()[:, :4] - g["logits_head"]).max() < 1e-3
    grads = eng.grad_dict()
    worst = TO.compare_grads(grads, g, "g/", rtol=1e-3)
    total = float(torch.sqrt(sum(v.double().pow(2).sum() for v in grads.values())))
    assert abs(total - float(g["g/total_norm"])) <= 1e-3 * float(g["g/total_norm"])
    print("worst relative gradient error vs the reference:", worst)


def test_staged_backward_is_the_whole_backward():
    g = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "vae_train.npz"))
    feat, units, lens = _batch(g)
    eng, _ = _engine("f32")
    noise = torch.from_numpy(g["post_noise"])
    eng.forward(feat, units, lens, noise=noise)
    eng.zero_grad()
    eng.backward()
    whole = eng.grads.clone()
    eng.forward(feat, units, lens, noise=noise)
    eng.zero_grad()
    for st in range(eng.n_stages):
        eng.backward(st, st)
        off, cnt = eng.stage_ranges()[st]
        assert torch.equal(eng.grads[off: off + cnt], whole[off: off + cnt]), f"stage {st} did not complete its range"
    assert torch.equal(eng.grads, whole)


def test_vae_gradients_bf16_follow_the_oracle(golden):
    g = golden("vae_train")
    feat, units, lens = _batch(g)
    eng, sd = _engine("bf16")
    stats = eng.forward(feat, units, lens, noise=torch.from_numpy(g["post_noise"]), ntokens=int(lens.sum()))
    eng.zero_grad()
    eng.backward()
    s = stats.cpu().double().numpy()
    for i, k in enumerate(("loss", "nll_loss", "mse_loss", "kl_loss")):
        assert abs(s[i] - float(g[k])) <= 2e-2 * max(1.0, abs(float(g[k]))), (k, s[i], float(g[k]))
    _, want = TO.vae_loss_and_grads(sd, CFG, feat, units, lens, torch.from_numpy(g["post_noise"]))
    got = eng.grad_dict()
    dot = sum((got[k].double() * want[k].double()).sum() for k in want)
    n1 = torch.sqrt(sum(got[k].double().pow(2).sum() for k in want))
    n2 = torch.sqrt(sum(want[k].double().pow(2).sum() for k in want))
    assert dot / (n1 * n2) > 0.999, float(dot / (n1 * n2))
    assert abs(float(n1 / n2) - 1) < 2e-2
    # every tensor individually: bf16 operand rounding relative to its own size, plus a floor relative to the whole gradient
    # (the q / k projections of a freshly initialised attention have gradients ~1e-4 of the total: pure rounding noise there)
    for k in want:
        err = float((got[k].double() - want[k].double()).norm())
        assert err <= 3e-2 * float(want[k].double().norm()) + 5e-4 * float(n2), (k, err, float(want[k].double().norm()), float(n2))


def test_five_update_trajectory_matches_reference_f32(golden):
    from diffnorm_amd import training

    g = golden("vae_train")
    feat, units, lens = _batch(g)
    lr, warm, warm_init, b1, b2, clip = (float(v) for v in g["hyper"])
    eng, _ = _engine("f32")
    tr = training.VaeTrainer(eng, lr=lr, betas=(b1, b2), clip_norm=clip, warmup_updates=int(warm), warmup_init_lr=warm_init,
                             attn_dropout=0.0)  # fixtures: reference in eval()
    sample = {"reduce_target": feat, "reduce_target_unit": units, "reduce_target_lengths": lens, "ntokens": int(lens.sum()),
              "nsentences": feat.shape[0]}
    traj = g["traj"]
    for it in range(traj.shape[0]):
        logged, norm = tr.train_step([sample], noises=[torch.from_numpy(g[f"traj_noise{it}"])])
        got = logged.cpu().double().numpy()
        for col, name in enumerate(("loss", "nll", "mse", "kl")):
            assert abs(got[col] - traj[it, col]) <= 2e-3 * max(1.0, abs(traj[it, col])), (it, name, got[col], traj[it, col])
        assert abs(got[4] - traj[it, 4]) < 1e-2
        assert abs(float(norm) - traj[it, 5]) <= 2e-3 * traj[it, 5], (it, float(norm), traj[it, 5])
        assert abs(tr.adam.get_lr() - traj[it, 6]) <= 1e-12 + 1e-9 * traj[it, 6]
    TO.compare_grads(eng.state_dict(), g, "p_end/", rtol=2e-3)


def _plugin_objects(dtype="f32"):
    import types

    from diffnorm_amd import fairseq_plugin  # noqa: F401  (registers the names)
    from diffnorm_amd.fairseq_plugin import registry

    args = types.SimpleNamespace(arch="speech_vae_decoder", criterion="speech_vae_decoder_loss", latent_dim=CFG.latent_dim,
                                 feature_dim=CFG.dim, hip_dtype=dtype, target_code_size=1000, data="")
    task = registry.TASK_REGISTRY["speech_decoder"].setup_task(args)
    model = task.build_model(args)
    sd = O.make_vae_state_dict(CFG, "train")
    model.load_state_dict({"encoder." + k: v for k, v in sd.items()}, strict=True)
    model.to(DEV)
    model.encoder.attn_dropout = 0.0  # the gradient fixtures were taken with the reference in eval() (oracle/gen_golden_train.py)
    return task, model, task.build_criterion(args)


def _sample(g, noise):
    feat, units, lens = _batch(g)
    return {"net_input": {"src_tokens": feat, "src_lengths": lens}, "reduce_target": feat, "reduce_target_unit": units,
            "reduce_target_lengths": lens, "target": feat, "target_unit": units, "target_lengths": lens, "ntokens": int(lens.sum()),
            "nsentences": feat.shape[0], "posterior_noise": noise}


def test_plugin_train_step_follows_the_reference_trajectory(golden):
    """task.train_step -> criterion -> model -> HIP engine, driven like fairseq's trainer (multiply_grads, clip_grad_norm, lr,
    step) through the FlatOptimizer: the same five updates as the reference's own Adam on its own modules."""
    from diffnorm_amd import optim

    g = golden("vae_train")
    lr, warm, warm_init, b1, b2, clip = (float(v) for v in g["hyper"])
    task, model, criterion = _plugin_objects("f32")
    eng = model.encoder.enable_training()
    assert [n for n, _ in model.named_parameters()] == ["encoder.flat_params"]
    assert set(model.state_dict()) == {"encoder." + k for k in O.make_vae_state_dict(CFG, "train")}
    opt = optim.FlatOptimizer(eng, lr=lr, betas=(b1, b2))
    sched = optim.InverseSquareRootSchedule(lr, int(warm), warm_init)
    traj = g["traj"]
    for it in range(traj.shape[0]):
        opt.zero_grad()
        loss, sample_size, log = task.train_step(_sample(g, torch.from_numpy(g[f"traj_noise{it}"])), model, criterion, opt, it)
        opt.multiply_grads(1.0 / sample_size)  # world / sample_size on one worker (trainer.py:918-933)
        norm = opt.clip_grad_norm(clip)
        opt.set_lr(sched.step_update(it))
        opt.step()
        for col, k in enumerate(("loss", "nll_loss", "mse_loss", "kl_loss")):
            assert abs(log[k] - traj[it, col]) <= 2e-3 * max(1.0, abs(traj[it, col])), (it, k, log[k], traj[it, col])
        assert abs(float(norm) - traj[it, 5]) <= 2e-3 * traj[it, 5]
        assert sample_size == 3 and abs(float(loss.detach()) - log["loss"]) < 1e-6
    TO.compare_grads({k[len("encoder."):]: v for k, v in model.state_dict().items()}, g, "p_end/", rtol=2e-3)
    # ignore_grad (fairseq's dummy batches): forward + logging, no gradient
    opt.zero_grad()
    task.train_step(_sample(g, torch.from_numpy(g["post_noise"])), model, criterion, opt, 5, ignore_grad=True)
    assert float(eng.grads.abs().max()) == 0.0


class _FairseqStyleOptimizer:
    """The slice of fairseq's FairseqOptimizer a train step touches, over a plain torch optimizer -- including its zero_grad, which
    sets `p.grad = None` (fairseq/optim/fairseq_optimizer.py:129-133) -- as the reference trainer builds it: from
    `model.parameters()`, after `model.to(device)`, before the first train_step (fairseq/trainer.py:292)."""

    def __init__(self, model, **kw):
        self._optimizer = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], **kw)

    @property
    def params(self):
        for g in self._optimizer.param_groups:
            yield from g["params"]

    def backward(self, loss):
        loss.backward()

    def multiply_grads(self, c):
        for p in self.params:
            if p.grad is not None:
                p.grad.mul_(c)

    def step(self):
        self._optimizer.step()

    def zero_grad(self):
        for p in self.params:
            p.grad = None


class _FairseqAdamThroughData(_FairseqStyleOptimizer):
    """The update of fairseq's own Adam restated (fairseq/optim/adam.py:185-236): it works on `p.data` -- `p_data_fp32 = p.data;
    exp_avg.mul_(beta1).add_(grad, alpha=1 - beta1); ...; p_data_fp32.addcdiv_(exp_avg, denom, value=-step_size)` -- and in-place
    operations on `.data` do NOT move the parameter's version counter, which is what round 3's bridge watched."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.98), eps=1e-8):
        self._params = [p for p in model.parameters() if p.requires_grad]
        self.lr, self.betas, self.eps, self.state = lr, betas, eps, {}

    @property
    def params(self):
        yield from self._params

    def step(self):
        for p in self._params:
            if p.grad is None:
                continue
            grad, p_data = p.grad.data.float(), p.data
            st = self.state.setdefault(p, {"step": 0, "exp_avg": torch.zeros_like(p_data), "exp_avg_sq": torch.zeros_like(p_data)})
            st["step"] += 1
            b1, b2 = self.betas
            st["exp_avg"].mul_(b1).add_(grad, alpha=1 - b1)
            st["exp_avg_sq"].mul_(b2).addcmul_(grad, grad, value=1 - b2)
            denom = st["exp_avg_sq"].sqrt().add_(self.eps)
            step_size = self.lr * math.sqrt(1 - b2 ** st["step"]) / (1 - b1 ** st["step"])
            p_data.addcdiv_(st["exp_avg"], denom, value=-step_size)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_level1_with_an_optimizer_that_updates_through_p_data(golden, dtype):
    """under `--optimizer adam` the reference's optimizer updates `flat_params` through `p.data`, the version
    counter stays put, and round 3's bridge never refreshed the bf16 working copy / the transposed weights -- bf16 training did
    nothing, f32 back-propagated through stale weights.  The bridge now keeps its own flag (latent_module._prepare_step): three
    updates by such an optimizer equal the HIP-backed FlatOptimizer's (same kernel contract, eps inside sqrt(v)), the version
    counter provably does not move, and the loss moves."""
    import types

    from diffnorm_amd import fairseq_plugin, optim  # noqa: F401
    from diffnorm_amd.fairseq_plugin import registry

    g = golden("vae_train")

    def build():
        args = types.SimpleNamespace(arch="speech_vae_decoder", criterion="speech_vae_decoder_loss", latent_dim=CFG.latent_dim,
                                     feature_dim=CFG.dim, hip_dtype=dtype, target_code_size=1000, data="", optimizer="adam", lr=[1e-3])
        task = registry.TASK_REGISTRY["speech_decoder"].setup_task(args)
        model = task.build_model(args)
        model.load_state_dict({"encoder." + k: v for k, v in O.make_vae_state_dict(CFG, "train").items()}, strict=True)
        model.to(DEV)
        model.encoder.attn_dropout = 0.0
        return task, model, task.build_criterion(args)

    task, model, criterion = build()
    ext = _FairseqAdamThroughData(model, lr=1e-3, betas=(0.9, 0.98), eps=1e-8)
    task2, model2, criterion2 = build()
    ref = optim.FlatOptimizer(model2.encoder._train_engine, lr=1e-3, betas=(0.9, 0.98), eps=1e-8)
    v0 = model.encoder.flat_params._version
    losses = []
    for it in range(3):
        sample = _sample(g, torch.from_numpy(g[f"traj_noise{it % 3}"]))
        ext.zero_grad()
        loss, n, _ = task.train_step(sample, model, criterion, ext, it)
        ref.zero_grad()
        loss2, _, _ = task2.train_step(sample, model2, criterion2, ref, it)
        losses.append(float(loss.detach()))
        tol = 1e-5 if dtype == "f32" else 2e-3
        assert abs(float(loss.detach()) - float(loss2.detach())) <= tol * abs(float(loss2.detach())), (it, float(loss.detach()), float(loss2.detach()))
        ext.multiply_grads(1.0 / n)
        ref.multiply_grads(1.0 / n)
        ext.step()
        ref.step()
        pa, pb = model.encoder._train_engine.master, model2.encoder._train_engine.master
        assert float((pa - pb).norm() / pb.norm()) < (1e-6 if dtype == "f32" else 1e-4), (it, float((pa - pb).norm() / pb.norm()))
    assert model.encoder.flat_params._version == v0, "the premise: an update through p.data is invisible to the version counter"
    # the engine really computes with the updated weights: the same batch again gives a different loss than before the updates
    sample = _sample(g, torch.from_numpy(g["traj_noise0"]))
    ext.zero_grad()
    loss_after, _, _ = task.train_step(sample, model, criterion, ext, 3)
    assert abs(float(loss_after.detach()) - losses[0]) > 1e-3 * abs(losses[0])
    eng = model.encoder._train_engine
    if eng.work is not eng.master:  # bf16: the working copy is the rounding of the master buffer as of the last forward
        assert torch.equal(eng.work, eng.master.to(torch.bfloat16))


def test_level1_reference_trainer_order_with_an_external_optimizer(golden):
    """The documented Level-1 path: plugin + the trainer's OWN optimizer.  A training namespace (--optimizer set) makes the model
    switch to the HIP training engine when it is moved to the GPU, so an optimizer built afterwards from `model.parameters()`
    -- and before any train_step -- holds `flat_params`; fairseq-style `zero_grad` (grad = None) clears the engine's gradient
    buffer and the backward restores the alias; an update by that optimizer is picked up by the next forward (bf16 working copy /
    transposed weights refreshed).  Three updates equal the HIP-backed FlatOptimizer's; a loss scale flows through
    `optimizer.backward(loss * scale)`; a model built without a training namespace refuses an optimizer that predates the switch."""
    import types

    from diffnorm_amd import fairseq_plugin, optim  # noqa: F401
    from diffnorm_amd.fairseq_plugin import registry

    g = golden("vae_train")

    def build(training_ns):
        args = types.SimpleNamespace(arch="speech_vae_decoder", criterion="speech_vae_decoder_loss", latent_dim=CFG.latent_dim,
                                     feature_dim=CFG.dim, hip_dtype="f32", target_code_size=1000, data="")
        if training_ns:
            args.optimizer, args.lr = "adam", [1e-3]
        task = registry.TASK_REGISTRY["speech_decoder"].setup_task(args)
        model = task.build_model(args)
        model.load_state_dict({"encoder." + k: v for k, v in O.make_vae_state_dict(CFG, "train").items()}, strict=True)
        model.to(DEV)
        model.encoder.attn_dropout = 0.0
        return task, model, task.build_criterion(args)

    task, model, criterion = build(True)
    assert [n for n, _ in model.named_parameters()] == ["encoder.flat_params"], "the switch happens with model.to(device)"
    # (eps far below every gradient: torch's Adam adds it to sqrt(v_hat), the fairseq-contract kernel to sqrt(v) before the bias
    # correction -- with 1e-8 the two differ visibly on near-zero gradients; the pads' 0 / (0 + eps) is 0 either way)
    ext = _FairseqStyleOptimizer(model, lr=1e-3, betas=(0.9, 0.98), eps=1e-15)  # BEFORE the first train_step
    task2, model2, criterion2 = build(True)
    ref = optim.FlatOptimizer(model2.encoder._train_engine, lr=1e-3, betas=(0.9, 0.98), eps=1e-15)
    for it in range(3):
        sample = _sample(g, torch.from_numpy(g[f"traj_noise{it}"]))
        ext.zero_grad()
        assert model.encoder.flat_params.grad is None
        loss, n, _ = task.train_step(sample, model, criterion, ext, it)
        assert model.encoder.flat_params.grad is model.encoder._train_engine.grads
        ref.zero_grad()
        loss2, _, _ = task2.train_step(sample, model2, criterion2, ref, it)
        assert abs(float(loss.detach()) - float(loss2.detach())) <= 1e-5 * abs(float(loss2.detach())), (it, float(loss.detach()), float(loss2.detach()))
        ga, gb = model.encoder._train_engine.grads, model2.encoder._train_engine.grads
        assert float((ga - gb).norm() / gb.norm()) < 1e-4, (it, float((ga - gb).norm() / gb.norm()))
        ext.multiply_grads(1.0 / n)
        ref.multiply_grads(1.0 / n)
        ext.step()
        ref.step()
        pa, pb = model.encoder._train_engine.master, model2.encoder._train_engine.master
        assert float((pa - pb).norm() / pb.norm()) < 1e-5, (it, float((pa - pb).norm() / pb.norm()))  # torch's Adam vs the fairseq-contract kernel
    # loss scaling: the bridge takes any scalar upstream gradient
    sample = _sample(g, torch.from_numpy(g["post_noise"]))
    ext.zero_grad()
    loss, _, _ = criterion(model, sample)
    ext.backward(loss)
    plain = model.encoder._train_engine.grads.clone()
    ext.zero_grad()
    loss, _, _ = criterion(model, sample)
    ext.backward(loss * 128.0)
    assert torch.equal(model.encoder._train_engine.grads / 128.0, plain)
    loss, _, _ = criterion(model, sample)  # a second micro-batch accumulates c * g on top (grad not None: no zeroing)
    ext.backward(loss * 128.0)
    assert float((model.encoder._train_engine.grads / 256.0 - plain).norm() / plain.norm()) < 1e-6
    # without a training namespace the model keeps its per-tensor parameters until train_step: an optimizer built before is refused
    task3, model3, criterion3 = build(False)
    assert len(list(model3.parameters())) > 1
    stale = _FairseqStyleOptimizer(model3, lr=1e-3)
    with pytest.raises(RuntimeError, match="flat_params"):
        task3.train_step(sample, model3, criterion3, stale, 0)


def test_reference_style_criterion_differentiates_the_model_outputs(golden):
    """A caller that builds its loss from the model's (mse_loss, lm_logits, kl_loss) -- the reference criterion's own code
    path -- gets the same parameter gradients as the fused criterion: autograd hands d loss / d logits to the HIP backward."""
    from diffnorm_amd.latent_module import label_smoothed_nll_loss, lengths_to_mask

    g = golden("vae_train")
    feat, units, lens = _batch(g)
    noise = torch.from_numpy(g["post_noise"])
    task, model, criterion = _plugin_objects("f32")
    eng = model.encoder.enable_training()
    eng.zero_grad()
    loss, _, _ = criterion(model, _sample(g, noise))
    loss.backward()
    fused = eng.grads.clone()
    eng.zero_grad()
    mse, logits, kl = model.encoder(feat, units, lengths_to_mask(lens, feat.shape[1]), noise=noise)
    lprobs = torch.log_softmax(logits, dim=-1).view(-1, logits.size(-1))
    tot, _ = label_smoothed_nll_loss(lprobs, units.to(DEV).view(-1), 0.1, ignore_index=0, reduce=True)
    own = 0.1 * tot / int(lens.sum()) + 10 * mse + 0.0001 * kl
    assert abs(float(own) - float(loss)) < 1e-4 * float(loss)
    own.backward()
    assert float((eng.grads - fused).norm() / fused.norm()) < 1e-5


def test_cascaded_vae_gradients_vs_oracle_autograd():
    """latent_dim = 16: three cascaded encoder / decoder WaveNets (the middle ones hand their gradient on in the arithmetic
    dtype) -- the HIP backward against torch autograd of the oracle's criterion, exact-fp32 mode."""
    from diffnorm_amd import training

    cfg = O.VaeConfig(dim=192, latent_dim=16)
    sd = O.make_vae_state_dict(cfg, "casc16")
    eng = training.VaeTrainEngine(sd, dim=cfg.dim, latent_dim=cfg.latent_dim, dtype="f32", device=DEV, depth=cfg.depth, heads=cfg.heads,
                                  dim_head=cfg.dim_head, stacks=cfg.stacks, layers=cfg.layers)
    feat = seeded((2, 36, cfg.dim), 71)
    lens = torch.tensor([36, 19])
    g = torch.Generator().manual_seed(72)
    units = torch.randint(4, 1004, (2, 36), generator=g).masked_fill(~O.lengths_to_mask(lens, 36), 0)
    noise = seeded((2, 36, cfg.z), 73)
    stats = eng.forward(feat, units, lens, noise=noise, ntokens=int(lens.sum()))
    eng.zero_grad()
    eng.backward()
    losses, want = TO.vae_loss_and_grads(sd, cfg, feat, units, lens, noise)
    assert abs(float(stats[0]) - losses["loss"]) < 1e-4 * abs(losses["loss"])
    got = eng.grad_dict()
    tot = float(torch.sqrt(sum(v.double().pow(2).sum() for v in want.values())))
    for k in want:
        err = float((got[k].double() - want[k].double()).norm())
        assert err <= 1e-3 * float(want[k].double().norm()) + 1e-6 * tot, (k, err)


def _eps_setup(dtype):
    from diffnorm_amd import training
    from gen_golden_configs import CHAIN_EPS

    vsd = O.make_vae_state_dict(CFG, "train")
    esd = O.make_eps_state_dict(CHAIN_EPS, "train")
    vae = training.VaeTrainEngine(vsd, dim=CFG.dim, latent_dim=CFG.latent_dim, dtype=dtype, device=DEV, depth=CFG.depth, heads=CFG.heads,
                                  dim_head=CFG.dim_head, stacks=CFG.stacks, layers=CFG.layers)
    eps = training.EpsTrainEngine(esd, CHAIN_EPS, vae, timesteps=200, dtype=dtype, device=DEV)
    return eps, vae, esd, vsd, CHAIN_EPS


def test_diffusion_loss_and_gradients_match_reference_f32(golden):
    """LatentDiscreteModel.forward (multitask) on the HIP diffusion training engine against the REAL reference's loss dict and
    autograd gradients of every eps-predictor parameter (tests/golden/eps_train.npz; the VAE is frozen)."""
    g = golden("eps_train")
    eps, vae, esd, vsd, ecfg = _eps_setup("f32")
    feat = seeded((3, 48, CFG.dim), 31)
    lens, units = torch.from_numpy(g["lens"]), torch.from_numpy(g["units"])
    T = lambda k: torch.from_numpy(g[k])
    z = O.vae_encode(vsd, CFG, feat, T("post_noise"))  # the frozen encoder's posterior sample (its parity: test_hip_engine)
    assert torch.equal(eps.state_dict()["wavenet.stacks.1.blocks.2.to_time_cond.weight"], esd["wavenet.stacks.1.blocks.2.to_time_cond.weight"])
    stats = eps.forward(feat, units, lens, z, T("times"), T("jitter"), T("true_noise"))
    vae.zero_grad()
    eps.zero_grad()
    eps.backward()
    s = stats.cpu().double().numpy()
    for i, k in enumerate(("total_loss", "nll_loss", "recon_mse_loss", "noise_loss", "acc")):
        ref = float(g["loss_" + k])
        assert abs(s[i] - ref) <= 2e-4 * max(1.0, abs(ref)), (k, s[i], ref)
    assert float(vae.grads.abs().max()) == 0.0  # frozen
    worst = TO.compare_grads(eps.grad_dict(), g, "g/", rtol=1e-3)
    print("diffusion training: worst relative gradient error vs the reference:", worst)
    # staged == whole, and every stage completes its range
    whole = eps.grads.clone()
    eps.forward(feat, units, lens, z, T("times"), T("jitter"), T("true_noise"))
    eps.zero_grad()
    for st in range(eps.n_stages):
        eps.backward(st, st)
        off, cnt = eps.stage_ranges()[st]
        assert torch.equal(eps.grads[off: off + cnt], whole[off: off + cnt]), st
    assert sum(c for _, c in eps.stage_ranges()) == eps.n_params


def test_diffusion_gradients_bf16_follow_the_oracle(golden):
    g = golden("eps_train")
    eps, vae, esd, vsd, ecfg = _eps_setup("bf16")
    feat = seeded((3, 48, CFG.dim), 31)
    lens, units = torch.from_numpy(g["lens"]), torch.from_numpy(g["units"])
    mask = O.lengths_to_mask(lens, 48)
    T = lambda k: torch.from_numpy(g[k])
    z = O.vae_encode(vsd, CFG, feat, T("post_noise"))
    stats = eps.forward(feat, units, lens, z, T("times"), T("jitter"), T("true_noise"))
    eps.zero_grad()
    eps.backward()
    for i, k in enumerate(("total_loss", "nll_loss", "recon_mse_loss", "noise_loss")):
        ref = float(g["loss_" + k])
        assert abs(float(stats[i]) - ref) <= 2e-2 * max(1.0, abs(ref)), (k, float(stats[i]), ref)
    _, want = TO.eps_loss_and_grads(esd, ecfg, vsd, CFG, 200, feat, units, mask, T("times"), T("post_noise"), T("jitter"), T("true_noise"))
    got = eps.grad_dict()
    dot = sum((got[k].double() * want[k].double()).sum() for k in want)
    n1 = torch.sqrt(sum(got[k].double().pow(2).sum() for k in want))
    n2 = torch.sqrt(sum(want[k].double().pow(2).sum() for k in want))
    print("bf16 diffusion gradient: cosine", float(dot / (n1 * n2)), "norm ratio", float(n1 / n2))
    assert dot / (n1 * n2) > 0.995 and abs(float(n1 / n2) - 1) < 5e-2


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_weight_gradients_on_the_side_stream_change_nothing(golden, dtype, hip_option):
    """The transformer layers' weight gradients run on a second stream beside the data-gradient chain (train_engine.hip: WgSide);
    DN_WGRAD_STREAM=0 keeps them on the caller's stream.  Same kernels, same reductions: the flat gradient buffers agree bit for
    bit, whole and staged backward, VAE and diffusion step, over repeated runs (a missing wait would show as a difference)."""
    g = golden("vae_train")
    feat, units, lens = _batch(g)
    noise = torch.from_numpy(g["post_noise"])
    eng, _ = _engine(dtype)
    grads = {}
    for mode in ("0", "1", "1", "0", "1"):
        hip_option("wgrad_stream", int(mode))
        eng.forward(feat, units, lens, noise=noise)
        eng.zero_grad()
        if mode == "1" and "staged" not in grads:
            for st in range(eng.n_stages):
                eng.backward(st, st)
            grads["staged"] = eng.grads.clone()
        else:
            eng.backward()
        torch.cuda.synchronize()
        grads.setdefault(mode, eng.grads.clone())
        assert torch.equal(eng.grads, grads["0"]), mode
    assert torch.equal(grads["staged"], grads["0"])
    ge = golden("eps_train")
    eps, vae, esd, vsd, ecfg = _eps_setup(dtype)
    T = lambda k: torch.from_numpy(ge[k])
    z = O.vae_encode(vsd, CFG, feat, T("post_noise"))
    ref = None
    for mode in ("0", "1", "1", "0", "1"):
        hip_option("wgrad_stream", int(mode))
        eps.forward(feat, torch.from_numpy(ge["units"]), torch.from_numpy(ge["lens"]), z, T("times"), T("jitter"), T("true_noise"))
        eps.zero_grad()
        eps.backward()
        torch.cuda.synchronize()
        ref = eps.grads.clone() if ref is None else ref
        assert torch.equal(eps.grads, ref), mode


def _grads_close(got, want, rtol):
    total = float(torch.sqrt(sum(v.double().pow(2).sum() for v in want.values())))
    worst = 0.0
    for k, w in want.items():
        err = float((got[k].double().cpu() - w.double()).abs().max())
        scale = max(float(w.abs().max()), 1e-3 * total / max(w.numel(), 1) ** 0.5)
        worst = max(worst, err / scale)
        assert err <= rtol * scale, (k, err, scale)
    return worst


def test_train_mode_attention_dropout_vae(golden):
    """Train mode (Attention(dropout=0.1), latent_module.py:338,668): the engine's counter-hash mask handed to the oracle's
    autograd gives the same losses and parameter gradients -- every layer's mask, forward and backward, is the one restated in
    oracle/dropout_mask.py; a second forward draws another mask; eval mode (p = 0) is the fixture case."""
    from dropout_mask import layer_keep

    g = golden("vae_train")
    feat, units, lens = _batch(g)
    noise = torch.from_numpy(g["post_noise"])
    eng, sd = _engine("f32")
    eng.attn_dropout, eng.dropout_seed = 0.1, 77
    stats = eng.forward(feat, units, lens, noise=noise, ntokens=int(lens.sum()))
    eng.zero_grad()
    eng.backward()
    lo, hi = eng._batch.dropout_seed_lo, eng._batch.dropout_seed_hi
    assert (lo, hi) != (0, 0)
    with O.attention_dropout("vae", 0.1, layer_keep(0.1, lo, hi)):
        want_loss, want = TO.vae_loss_and_grads(sd, CFG, feat, units, lens, noise)
    s = stats.cpu().double().numpy()
    for i, k in enumerate(("loss", "nll_loss", "mse_loss", "kl_loss")):
        assert abs(s[i] - want_loss[k]) <= 1e-4 * max(1.0, abs(want_loss[k])), (k, s[i], want_loss[k])
    assert abs(s[0] - float(g["loss"])) > 1e-4  # not the eval-mode loss
    print("train-mode VAE: worst relative gradient error vs the oracle with the same mask:", _grads_close(eng.grad_dict(), want, 1e-3))
    again = eng.forward(feat, units, lens, noise=noise, ntokens=int(lens.sum()))
    assert (eng._batch.dropout_seed_lo, eng._batch.dropout_seed_hi) != (lo, hi) and float(again[0]) != float(stats[0])
    eng.attn_dropout = 0.0
    evald = eng.forward(feat, units, lens, noise=noise, ntokens=int(lens.sum()))
    assert abs(float(evald[0]) - float(g["loss"])) <= 1e-4 * float(g["loss"])


def test_train_mode_attention_dropout_diffusion(golden):
    """The eps-predictor in train mode, the frozen VAE in eval mode (latent_module.py:1530) -- against the oracle with the mask."""
    from dropout_mask import layer_keep

    g = golden("eps_train")
    eps, vae, esd, vsd, ecfg = _eps_setup("f32")
    feat = seeded((3, 48, CFG.dim), 31)
    lens, units = torch.from_numpy(g["lens"]), torch.from_numpy(g["units"])
    mask = O.lengths_to_mask(lens, 48)
    T = lambda k: torch.from_numpy(g[k])
    z = O.vae_encode(vsd, CFG, feat, T("post_noise"))
    eps.attn_dropout, eps.dropout_seed = 0.1, 5
    stats = eps.forward(feat, units, lens, z, T("times"), T("jitter"), T("true_noise"))
    eps.zero_grad()
    eps.backward()
    lo, hi = eps._batch.dropout_seed_lo, eps._batch.dropout_seed_hi
    with O.attention_dropout("eps", 0.1, layer_keep(0.1, lo, hi)):
        want_loss, want = TO.eps_loss_and_grads(esd, ecfg, vsd, CFG, 200, feat, units, mask, T("times"), T("post_noise"), T("jitter"),
                                                T("true_noise"))
    for i, k in enumerate(("total_loss", "nll_loss", "recon_mse_loss", "noise_loss")):
        assert abs(float(stats[i]) - want_loss[k]) <= 2e-4 * max(1.0, abs(want_loss[k])), (k, float(stats[i]), want_loss[k])
    assert abs(float(stats[0]) - float(g["loss_total_loss"])) > 1e-4
    print("train-mode diffusion: worst relative gradient error vs the oracle with the same mask:", _grads_close(eps.grad_dict(), want, 1e-3))


def test_module_train_eval_modes_switch_dropout(golden):
    """nn.Module semantics: model.train() turns the attention dropout on (seeded off torch's generator, so a re-seeded step
    repeats), model.eval() turns it off."""
    g = golden("vae_train")
    task, model, criterion = _plugin_objects("f32")
    model.encoder.attn_dropout = 0.1
    model.encoder.enable_training()
    sample = _sample(g, torch.from_numpy(g["post_noise"]))
    model.eval()
    with torch.no_grad():
        ev, _, _ = criterion(model, sample)
        ev2, _, _ = criterion(model, sample)
    assert float(ev) == float(ev2)  # eval: no mask, nothing drawn
    model.train()
    torch.manual_seed(11)
    a, _, _ = criterion(model, sample)
    torch.manual_seed(11)
    b, _, _ = criterion(model, sample)
    c, _, _ = criterion(model, sample)
    assert float(a) == float(b) and float(a) != float(c) and float(a) != float(ev)


def test_plugin_diffusion_train_step_matches_reference(golden):
    """--task speech_diffusion_discrete --criterion ddpm_discrete_loss --arch diff_discrete: task.train_step through the plugin
    (criterion -> DiffDiscreteModel -> LatentDiscreteModel.forward -> HIP diffusion engine, loss.backward() = the HIP backward)
    gives the reference's loss dict and gradients; the checkpoint keeps the reference's 516-tensor layout."""
    import types

    from diffnorm_amd import fairseq_plugin, optim  # noqa: F401
    from diffnorm_amd.fairseq_plugin import registry
    from gen_golden_configs import CHAIN_EPS

    g = golden("eps_train")
    args = types.SimpleNamespace(arch="diff_discrete", criterion="ddpm_discrete_loss", latent_dim=CFG.latent_dim, feature_dim=CFG.dim,
                                 denoiser_dim=CHAIN_EPS.dim, hip_dtype="f32", multitask=True, diffusion_timesteps=200, speech_decoder_ckpt=None,
                                 target_code_size=1000, data="")
    task = registry.TASK_REGISTRY["speech_diffusion_discrete"].setup_task(args)
    model = task.build_model(args)
    vsd, esd = O.make_vae_state_dict(CFG, "train"), O.make_eps_state_dict(CHAIN_EPS, "train")
    full = {"encoder.speech_decoder." + k: v for k, v in vsd.items()}
    full.update({"encoder.model." + k: v for k, v in esd.items()})
    full["encoder.model.pos_embed._float_tensor"] = torch.zeros(1)
    model.load_state_dict(full, strict=True)
    model.to(DEV)
    criterion = task.build_criterion(args)
    model.encoder.attn_dropout = 0.0  # fixtures: reference in eval()
    eng = model.encoder.enable_training()
    assert set(model.state_dict()) == set(full)
    assert [n for n, p in model.named_parameters() if p.requires_grad] == ["encoder.model.flat_params"]
    feat, units, lens = _batch(g)
    T = lambda k: torch.from_numpy(g[k])
    sample = {"net_input": {"src_tokens": feat, "src_lengths": lens}, "reduce_target": feat, "reduce_target_unit": units,
              "reduce_target_lengths": lens, "ntokens": int(lens.sum()), "nsentences": 3,
              "diffusion_draws": {"times": T("times"), "post_noise": T("post_noise"), "jitter_noise": T("jitter"), "true_noise": T("true_noise")}}
    opt = optim.FlatOptimizer(eng, lr=1e-4, betas=(0.9, 0.98))
    opt.zero_grad()
    loss, sample_size, log = task.train_step(sample, model, criterion, opt, 0)
    for k, ref_k in (("loss", "total_loss"), ("nll_loss", "nll_loss"), ("mse_loss", "recon_mse_loss"), ("noise_loss", "noise_loss")):
        ref = float(g["loss_" + ref_k])
        assert abs(log[k] - ref) <= 1e-3 * max(1.0, abs(ref)), (k, log[k], ref)
    TO.compare_grads(eng.grad_dict(), g, "g/", rtol=1e-3)
    before = model.state_dict()["encoder.model.final_proj.weight"].clone()
    opt.multiply_grads(1.0 / sample_size)
    opt.clip_grad_norm(2.0)
    opt.step()
    after = model.state_dict()
    assert not torch.equal(after["encoder.model.final_proj.weight"], before)
    assert torch.equal(after["encoder.speech_decoder.decoder_lm.weight"].cpu(), vsd["decoder_lm.weight"])  # frozen
    # sampling still works on the updated weights (the inference engine is rebuilt from the master buffer)
    mask = O.lengths_to_mask(lens, 48).to(DEV)
    toks, _, total, _ = model.encoder.ddim_sample(feat.to(DEV), input_mask=mask, ref_units=(units - 4).to(DEV), start_step=3)
    assert total == int(lens.sum()) and [t.shape[0] for t in toks] == lens.tolist()


def _cosine(got, want_samples, golden, prefix="g/"):
    """Cosine / norm ratio of a gradient against the fixture's strided samples (the fixture keeps samples, not whole tensors)."""
    dot = n1 = n2 = 0.0
    for name in [str(n) for n in golden[prefix + "names"]]:
        g = got[name].detach().double().cpu().flatten()
        key = f"{prefix}full/{name}" if f"{prefix}full/{name}" in golden else f"{prefix}samp/{name}"
        ref = torch.from_numpy(golden[key]).double().flatten()
        if key.startswith(prefix + "samp/"):
            g = g[:: (g.numel() + ref.numel() - 1) // ref.numel()]
        dot += float((g * ref).sum())
        n1 += float(g.pow(2).sum())
        n2 += float(ref.pow(2).sum())
    return dot / (n1 * n2) ** 0.5, (n1 / n2) ** 0.5


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_fullsize_vae_training_step_matches_reference(golden, dtype):
    """The RECIPE-sized VAE (dim 768, 8 x 96 heads, FFN inner 2048, 138.6 M parameters) at B = 2, T = 64: loss terms and the
    gradient of every parameter against the real reference's autograd (tests/golden/vae_train_full.npz: checksums + strided
    samples) -- the d_h = 96 attention backward, the 2048-wide GEGLU / causal-conv gradients and the 1004-way head at the sizes
    the recipe trains.  f32: every tensor within 1e-3; bf16: cosine / norm of the whole gradient."""
    from diffnorm_amd import training
    from gen_golden_configs import FULL_VAE

    g = golden("vae_train_full")
    sd = O.make_vae_state_dict(FULL_VAE, "full")
    eng = training.VaeTrainEngine(sd, dtype=dtype, device=DEV)
    del sd
    feat = seeded((2, 64, FULL_VAE.dim), 41)
    units, lens = torch.from_numpy(g["units"]), torch.from_numpy(g["lens"])
    stats, logits, _ = eng.forward(feat, units, lens, noise=torch.from_numpy(g["post_noise"]), ntokens=int(lens.sum()), want_logits=True)
    eng.zero_grad()
    eng.backward()
    s = stats.cpu().double().numpy()
    tol = 2e-4 if dtype == "f32" else 3e-2
    for i, k in enumerate(("loss", "nll_loss", "mse_loss", "kl_loss")):
        assert abs(s[i] - float(g[k])) <= tol * max(1.0, abs(float(g[k]))), (k, s[i], float(g[k]))
    grads = eng.grad_dict()
    if dtype == "f32":
        assert np.abs(logits.cpu().numpy()[:, :4, :64] - g["logits_head"]).max() < 1e-3
        worst = TO.compare_grads(grads, g, "g/", rtol=1e-3)
        print("full-size VAE: worst relative gradient error vs the reference:", worst)
    else:
        cos, ratio = _cosine(grads, None, g)
        print(f"full-size VAE bf16 gradient: cosine {cos:.5f}, norm ratio {ratio:.4f}")
        assert cos > 0.995 and abs(ratio - 1) < 5e-2
        _assert_bf16_gradient_per_tensor(_per_tensor_cosines(grads, g), "full-size VAE bf16 (B = 2, T = 64)")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_fullsize_diffusion_training_step_matches_reference(golden, dtype):
    """The RECIPE-sized eps-predictor (dim 512, depth 12, FFN inner 1365 padded to 1408, the 57 k-wide conditioning projection,
    260.6 M trained parameters) through the frozen recipe-sized VAE at B = 2, T = 64: LatentDiscreteModel.forward's loss dict and
    the gradient of every eps-predictor parameter against the real reference (tests/golden/eps_train_full.npz)."""
    from diffnorm_amd import training
    from gen_golden_configs import FULL_EPS, FULL_VAE

    g = golden("eps_train_full")
    vsd, esd = O.make_vae_state_dict(FULL_VAE, "full"), O.make_eps_state_dict(FULL_EPS, "full")
    vae = training.VaeTrainEngine(vsd, dtype=dtype, device=DEV)
    eps = training.EpsTrainEngine(esd, FULL_EPS, vae, timesteps=200, dtype=dtype, device=DEV)
    feat = seeded((2, 64, FULL_VAE.dim), 41)
    units, lens = torch.from_numpy(g["units"]), torch.from_numpy(g["lens"])
    T = lambda k: torch.from_numpy(g[k])
    # the frozen encoder's posterior sample: exact-fp32 inference engine in both modes (its parity: tests/test_hip_engine.py), so the
    # training engine under test sees the reference's z
    from diffnorm_amd import engine

    ve = engine.VaeEngine(vsd, dtype="f32", device=DEV)
    z = ve.sample_posterior(ve.encode_params(feat.to(DEV)), T("post_noise"))
    del ve, vsd, esd
    stats = eps.forward(feat, units, lens, z, T("times"), T("jitter"), T("true_noise"))
    vae.zero_grad()
    eps.zero_grad()
    eps.backward()
    s = stats.cpu().double().numpy()
    tol = 2e-4 if dtype == "f32" else 3e-2
    for i, k in enumerate(("total_loss", "nll_loss", "recon_mse_loss", "noise_loss")):
        ref = float(g["loss_" + k])
        assert abs(s[i] - ref) <= tol * max(1.0, abs(ref)), (k, s[i], ref)
    assert float(vae.grads.abs().max()) == 0.0  # frozen
    grads = eps.grad_dict()
    if dtype == "f32":
        worst = TO.compare_grads(grads, g, "g/", rtol=1e-3)
        print("full-size diffusion training: worst relative gradient error vs the reference:", worst)
    else:
        cos, ratio = _cosine(grads, None, g)
        print(f"full-size diffusion bf16 gradient: cosine {cos:.5f}, norm ratio {ratio:.4f}")
        assert cos > 0.99 and abs(ratio - 1) < 8e-2
        _assert_bf16_gradient_per_tensor(_per_tensor_cosines(grads, g), "full-size diffusion bf16 (B = 2, T = 64)")



def _per_tensor_cosines(got, golden, prefix="g/"):
    """Every tensor of the fixture on its own: (name, cosine against the fixture's whole tensor / strided sample, the reference
    tensor's l2 norm as a fraction of the whole gradient's norm, the absolute error of the sample as a fraction of that norm)."""
    names = [str(n) for n in golden[prefix + "names"]]
    total = float(golden[prefix + "total_norm"]) if prefix + "total_norm" in golden else float(
        np.sqrt(sum(float(golden[f"{prefix}chk/{n}"][1]) ** 2 for n in names)))
    rows = []
    for name in names:
        g = got[name].detach().double().cpu().flatten()
        key = f"{prefix}full/{name}" if f"{prefix}full/{name}" in golden else f"{prefix}samp/{name}"
        ref = torch.from_numpy(golden[key]).double().flatten()
        if key.startswith(prefix + "samp/"):
            g = g[:: (g.numel() + ref.numel() - 1) // ref.numel()]
        scale = (got[name].numel() / ref.numel()) ** 0.5  # a strided sample carries 1 / stride of the tensor's energy
        frac = float(golden[f"{prefix}chk/{name}"][1]) / total
        if float(ref.norm()) * scale < 0.1 * frac * total:
            # the sample misses the tensor's energy (a conv weight [cout, cin, 3] sampled with a stride that is a multiple of 3 sees one tap
            # only -- at T = 64 the tap of a dilation-64 block that reads nothing but the zero padding): no direction to compare, the
            # absolute-error rule below applies
            frac = 0.0
        cos = float((g * ref).sum() / max(float(g.norm() * ref.norm()), 1e-300))
        rows.append((name, cos, frac, float((g - ref).norm()) * scale / total))
    return rows


def _assert_bf16_gradient_per_tensor(rows, what, min_cos=0.99, floor=2e-3):
    """bf16 training against the reference's fp32 gradients, TENSOR BY TENSOR (round 3 checked one global cosine, which a wrong bias
    or gamma cannot move): cosine >= min_cos for every tensor that carries more than `floor` of the whole gradient's norm; a
    smaller one (the q / k projections of a freshly initialised attention, biases behind a norm: ~1e-4 of the total, where the
    2^-9 operand rounding of its producers is the signal's own size) must instead be wrong by no more than `floor` / 4 of the whole
    gradient's norm in absolute terms."""
    bad = []
    for name, cos, frac, aerr in rows:
        if frac > floor:
            if cos < min_cos:
                bad.append((name, f"cos {cos:.4f}", f"norm fraction {frac:.2e}"))
        elif aerr > floor / 4:
            bad.append((name, f"abs err {aerr:.2e} of the total norm", f"norm fraction {frac:.2e}"))
    worst = sorted((r for r in rows if r[2] > floor), key=lambda r: r[1])[:5]
    print(f"{what}: {len(rows)} tensors, {sum(r[2] > floor for r in rows)} above the floor; lowest cosines: " +
          ", ".join(f"{n} {c:.4f}" for n, c, _, _ in worst))
    assert not bad, bad[:10]


def _bench_batch(B, T, dim, seed):
    """oracle/gen_golden_train.py::bench_batch, regenerated (the fixture stores lengths and units; the features come from the seed)."""
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(64, T + 1, (B,), generator=g).sort(descending=True).values
    lens[0] = T
    mask = O.lengths_to_mask(lens, T)
    return seeded((B, T, dim), seed + 1) * mask.unsqueeze(-1), lens


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_bench_shape_vae_training_step_matches_reference(golden, dtype):
    """The RECIPE-sized VAE at the batch shape bench.py's VAE leg times (--max-tokens 15000 -> B = 24, T = 512, ragged lengths
    U[64, 512], 7.4 k valid frames): losses and the gradient of every parameter against the real reference's autograd
    (tests/golden/vae_train_batch.npz, oracle/gen_golden_train.py --full-batch) -- 12 k-frame weight gradients, K-sliced partial
    sums, the grouped WaveNet launches, tiles with sequence starts inside.  f32: every tensor within 1e-3; bf16: every tensor."""
    from diffnorm_amd import training
    from gen_golden_configs import FULL_VAE

    g = golden("vae_train_batch")
    eng = training.VaeTrainEngine(O.make_vae_state_dict(FULL_VAE, "full"), dtype=dtype, device=DEV)
    feat, lens = _bench_batch(24, 512, FULL_VAE.dim, int(g["batch_seed"]))
    assert torch.equal(lens, torch.from_numpy(g["lens"]))
    units = torch.from_numpy(g["units"])
    noise = seeded(tuple(int(v) for v in g["post_noise_shape"]), int(g["post_noise_seed"])).transpose(1, 2).contiguous()  # drawn [B, z, T]
    stats, logits, _ = eng.forward(feat, units, lens, noise=noise, ntokens=int(lens.sum()), want_logits=True)
    eng.zero_grad()
    eng.backward()
    s = stats.cpu().double().numpy()
    tol = 2e-4 if dtype == "f32" else 3e-2
    for i, k in enumerate(("loss", "nll_loss", "mse_loss", "kl_loss")):
        assert abs(s[i] - float(g[k])) <= tol * max(1.0, abs(float(g[k]))), (k, s[i], float(g[k]))
    grads = eng.grad_dict()
    if dtype == "f32":
        assert np.abs(logits.cpu().numpy()[:, :4, :64] - g["logits_head"]).max() < 1e-3
        worst = TO.compare_grads(grads, g, "g/", rtol=1e-3)
        print("bench-shape VAE: worst relative gradient error vs the reference:", worst)
    else:
        _assert_bf16_gradient_per_tensor(_per_tensor_cosines(grads, g), "bench-shape VAE bf16")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_bench_shape_diffusion_training_step_matches_reference(golden, dtype):
    """The RECIPE-sized eps-predictor through the frozen VAE at the batch shape bench.py's diffusion leg times (--max-tokens 12000 ->
    B = 16, T = 512, ragged): LatentDiscreteModel.forward's loss dict and the gradient of every eps-predictor parameter against the
    real reference (tests/golden/eps_train_batch.npz); the reference's four draws were INJECTED from portable seeds and are
    regenerated here."""
    from diffnorm_amd import engine, training
    from gen_golden_configs import FULL_EPS, FULL_VAE

    g = golden("eps_train_batch")
    vsd, esd = O.make_vae_state_dict(FULL_VAE, "full"), O.make_eps_state_dict(FULL_EPS, "full")
    vae = training.VaeTrainEngine(vsd, dtype=dtype, device=DEV)
    eps = training.EpsTrainEngine(esd, FULL_EPS, vae, timesteps=200, dtype=dtype, device=DEV)
    feat, lens = _bench_batch(16, 512, FULL_VAE.dim, int(g["batch_seed"]))
    assert torch.equal(lens, torch.from_numpy(g["lens"]))
    units = torch.from_numpy(g["units"])
    seeds, shapes = [int(v) for v in g["draw_seeds"]], [tuple(int(x) for x in row if x) for row in g["draw_shapes"]]
    times = torch.randint(1, 200, shapes[0], generator=torch.Generator().manual_seed(seeds[0]))
    assert torch.equal(times, torch.from_numpy(g["times"]))
    post = seeded(shapes[1], seeds[1]).transpose(1, 2).contiguous()  # drawn [B, z, T] (distributions.py:38)
    jitter, true_noise = seeded(shapes[2], seeds[2]), seeded(shapes[3], seeds[3])
    ve = engine.VaeEngine(vsd, dtype="f32", device=DEV)  # the frozen encoder's posterior sample: exact fp32, so the engine under test sees the reference's z
    z = ve.sample_posterior(ve.encode_params(feat.to(DEV)), post)
    del ve, vsd, esd
    stats = eps.forward(feat, units, lens, z, times, jitter, true_noise)
    vae.zero_grad()
    eps.zero_grad()
    eps.backward()
    s = stats.cpu().double().numpy()
    tol = 2e-4 if dtype == "f32" else 3e-2
    for i, k in enumerate(("total_loss", "nll_loss", "recon_mse_loss", "noise_loss")):
        ref = float(g["loss_" + k])
        assert abs(s[i] - ref) <= tol * max(1.0, abs(ref)), (k, s[i], ref)
    assert float(vae.grads.abs().max()) == 0.0  # frozen
    grads = eps.grad_dict()
    if dtype == "f32":
        worst = TO.compare_grads(grads, g, "g/", rtol=1e-3)
        print("bench-shape diffusion training: worst relative gradient error vs the reference:", worst)
    else:
        _assert_bf16_gradient_per_tensor(_per_tensor_cosines(grads, g), "bench-shape diffusion bf16")
