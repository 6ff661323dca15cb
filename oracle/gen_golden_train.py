"""TEST INFRASTRUCTURE ONLY.  Gradient / training-trajectory fixtures (SURVEY 8c fixture plan item 5, scope rows f2 and
(e)-training), produced in the build container by the REAL reference modules (loaded where they lie by oracle/ref_loader.py):

  vae_train.npz   SpeechVAEEncoderDecoder.forward (latent_module.py:1118-1142) + the criterion algebra of
                  fairseq/criterions/speech_vae_decoder_loss.py:45-95 on a small VAE; autograd gradients of every one of its
                  parameters (checksums of all, whole tensors of the small ones, strided samples of the large ones) and a
                  5-update trajectory driven like fairseq's trainer (fairseq/trainer.py:912-939: multiply_grads(world /
                  sample_size), clip_grad_norm(2.0), inverse_sqrt lr, the reference's own Adam class).
  eps_train.npz   LatentDiscreteModel.forward (latent_module.py:1514-1613, multitask) on a small VAE + eps-predictor pair with
                  the frozen VAE (diff_discrete.py:70-85): gradients of the eps-predictor's parameters, same storage.

The reference's attention dropout (p = 0.1, latent_module.py:338,668; the only train/eval difference on the path) is not
re-drawable from outside, so the fixtures are taken with the modules in eval() -- gradients flow identically, dropout is the
identity.  Random tensors the reference draws (posterior noise, t, jitter, target noise) are recorded and stored.

  vae_train_full.npz / eps_train_full.npz (--full): the same losses and gradients on the RECIPE-sized models (dim 768 / d_h 96 /
                  inner 2048; dim 512 / depth 12 / inner 1365 padded to 1408 / the 57 k-wide conditioning projection) at B = 2, T = 64:
                  checksums + strided samples only.

  vae_train_batch.npz / eps_train_batch.npz (--full-batch): the RECIPE-sized models at the batch shapes bench.py's training
                  legs actually time (scripts/vae/train.sh:5-9 --max-tokens 15000 -> B = 24 x T = 512; scripts/diffusion/train.sh
                  --max-tokens 12000 -> B = 16 x T = 512; ragged lengths U[64, 512], longest first, zero-padded features): 8-12 k
                  frames into every weight gradient, K-sliced partial sums, several slices -- checksums + strided samples only.
                  The reference's random draws are not recorded here (a [24, 512, 128] noise tensor is 6 MB) but INJECTED: while
                  the reference runs, torch.randn / randn_like / randint return portable seeded tensors (gen_golden_configs.seeded
                  with the seeds stored in the fixture), which the test regenerates.

Usage:  python oracle/gen_golden_train.py [--full | --full-batch]   ->  tests/golden/{vae_train,eps_train}[_full|_batch].npz
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import diffnorm_oracle as O  # noqa: E402
import ref_loader  # noqa: E402
from gen_golden import record_draws, ref_vae, save  # noqa: E402
from gen_golden_configs import CHAIN_EPS, CHAIN_VAE, seeded  # noqa: E402

SMALL = 20000       # tensors up to this many elements are stored whole
SAMPLES = 2048      # strided samples of the larger ones
UPDATES, LR, WARMUP, WARMUP_INIT, BETAS, CLIP = 5, 5e-4, 3, 1e-7, (0.9, 0.98), 2.0


def probe_vector(n: int) -> torch.Tensor:
    """Fixed +-1 pattern for the dot-product checksum (portable: a linear congruence, no RNG state)."""
    i = torch.arange(n, dtype=torch.int64)
    return (((i * 2654435761 + 12345) >> 7) & 1).to(torch.float64) * 2 - 1


def grad_record(prefix, named_grads, out, small=SMALL, samples=SAMPLES):
    """Per tensor: [sum, l2 norm, dot with probe_vector] in float64, plus the tensor itself (small) or a strided sample."""
    names = []
    for name, g in named_grads:
        g64 = g.detach().double().flatten()
        n = g64.numel()
        out[f"{prefix}chk/{name}"] = np.array([g64.sum().item(), g64.norm().item(), (g64 * probe_vector(n)).sum().item()])
        if n <= small:
            out[f"{prefix}full/{name}"] = g.detach().float().numpy()
        else:
            stride = (n + samples - 1) // samples
            samp = g.detach().float().flatten()[::stride]
            assert (n + samp.numel() - 1) // samp.numel() == stride, (name, n, stride)  # train_oracle.compare_grads re-derives the stride
            out[f"{prefix}samp/{name}"] = samp.numpy()
        names.append(name)
    out[prefix + "names"] = np.array(names)
    total = torch.sqrt(sum(g.detach().double().pow(2).sum() for _, g in named_grads))
    out[prefix + "total_norm"] = np.float64(total.item())


def vae_criterion_loss(lm_mod, vae, feat, units, mask, lens):
    """speech_vae_decoder_loss.py:60-83 on the reference model's outputs (the criterion file itself drags omegaconf)."""
    ls = sys.modules["fairseq.criterions.label_smoothed_cross_entropy"].label_smoothed_nll_loss
    mse, logits, kl = vae(feat, units, mask)
    lprobs = torch.log_softmax(logits, dim=-1).view(-1, logits.size(-1))
    target = units.view(-1)
    keep = target.ne(0)
    acc = torch.sum(lprobs.argmax(1).masked_select(keep).eq(target.masked_select(keep))) / torch.sum(keep)
    loss, nll = ls(lprobs, target, 0.1, ignore_index=0, reduce=True)
    ntokens = int(lens.sum())
    loss, nll = loss / ntokens, nll / ntokens
    return 0.1 * loss + 10 * mse + 0.0001 * kl, dict(nll_loss=nll, mse_loss=mse, kl_loss=kl, acc=acc), logits


def batch(vcfg):
    B, T = 3, 48
    feat = seeded((B, T, vcfg.dim), 31)
    lens = torch.tensor([48, 29, 40])
    mask = O.lengths_to_mask(lens, T)
    g = torch.Generator().manual_seed(32)
    units = torch.randint(4, 1004, (B, T), generator=g).masked_fill(~mask, 0)
    return feat, lens, mask, units


def gen_vae_train(lm):
    vcfg = CHAIN_VAE
    vsd = O.make_vae_state_dict(vcfg, "train")
    vae = ref_vae(lm, vcfg, vsd)  # eval(): dropout off, gradients on
    feat, lens, mask, units = batch(vcfg)
    out = dict(lens=lens, units=units)
    torch.manual_seed(501)
    with record_draws() as rec:
        loss, parts, logits = vae_criterion_loss(lm, vae, feat, units, mask, lens)
    assert len(rec.draws) == 1
    out["post_noise"] = rec.draws[0].transpose(1, 2).contiguous()  # [B,T,z]
    loss.backward()
    out["loss"] = loss.detach()
    out.update({k: v.detach() for k, v in parts.items()})
    out["logits_head"] = logits.detach()[:, :4]
    grad_record("g/", [(n, p.grad) for n, p in vae.named_parameters()], out)
    # trajectory, driven like fairseq's trainer on one worker (trainer.py:912-939): grads * (world / sample_size) with
    # sample_size = nsentences (speech_vae_decoder_loss.py:84), clip_grad_norm_(2.0), lr = inverse_sqrt(num_updates), Adam
    Adam, clip_grad_norm_, InverseSquareRootSchedule = ref_loader.load_reference_optim()
    params = [p for p in vae.parameters()]
    opt = Adam(params, lr=LR, betas=BETAS, eps=1e-8, weight_decay=0.0)
    holder = types.SimpleNamespace()
    holder.set_lr = lambda lr: [g.__setitem__("lr", lr) for g in opt.param_groups]
    holder.get_lr = lambda: opt.param_groups[0]["lr"]
    sched = InverseSquareRootSchedule(types.SimpleNamespace(lr=[LR], warmup_updates=WARMUP, warmup_init_lr=WARMUP_INIT), holder)
    nsent = feat.shape[0]
    traj = []
    for it in range(UPDATES):
        opt.zero_grad()
        torch.manual_seed(600 + it)
        with record_draws() as rec:
            loss, parts, _ = vae_criterion_loss(lm, vae, feat, units, mask, lens)
        out[f"traj_noise{it}"] = rec.draws[0].transpose(1, 2).contiguous()
        loss.backward()
        for p in params:
            p.grad.mul_(1.0 / nsent)
        norm = clip_grad_norm_(params, CLIP)
        lr = sched.step_update(it)
        opt.step()
        traj.append([loss.item(), parts["nll_loss"].item(), parts["mse_loss"].item(), parts["kl_loss"].item(), parts["acc"].item(),
                     norm.item(), lr])
    out["traj"] = np.array(traj, dtype=np.float64)  # columns: loss nll mse kl acc grad_norm lr
    out["hyper"] = np.array([LR, WARMUP, WARMUP_INIT, BETAS[0], BETAS[1], CLIP])
    # parameters after the trajectory: checksums (same record format)
    grad_record("p_end/", [(n, p.data) for n, p in vae.named_parameters()], out)
    save("vae_train", **out)


def gen_eps_train(lm):
    ecfg, vcfg = CHAIN_EPS, CHAIN_VAE
    esd = O.make_eps_state_dict(ecfg, "train")
    vsd = O.make_vae_state_dict(vcfg, "train")
    vae = ref_vae(lm, vcfg, vsd)
    for p in vae.parameters():
        p.requires_grad = False  # diff_discrete.py:79-82
    ldm = lm.LatentDiscreteModel(types.SimpleNamespace(encoder=vae), ecfg.dim, vcfg.z, timesteps=200, multitask=True)
    full = dict(esd)
    full["pos_embed._float_tensor"] = torch.zeros(1)
    ldm.model.load_state_dict(full, strict=True)
    ldm.eval()
    feat, lens, mask, units = batch(vcfg)
    out = dict(lens=lens, units=units)
    torch.manual_seed(777)
    with record_draws() as rec:
        ld = ldm(feat, units, tgt_mask=mask)
    assert len(rec.draws) == 4, len(rec.draws)
    out.update(times=rec.draws[0], post_noise=rec.draws[1].transpose(1, 2).contiguous(), jitter=rec.draws[2].contiguous(),
               true_noise=rec.draws[3].contiguous(), **{"loss_" + k: v.detach() for k, v in ld.items()})
    ld["total_loss"].backward()
    grad_record("g/", [(n, p.grad) for n, p in ldm.model.named_parameters() if p.grad is not None], out)
    out["no_grad_names"] = np.array([n for n, p in ldm.model.named_parameters() if p.grad is None])
    save("eps_train", **out)


def full_batch(dim):
    """B = 2, T = 64 at the recipe's sizes: one full-length and one ragged utterance."""
    B, T = 2, 64
    feat = seeded((B, T, dim), 41)
    lens = torch.tensor([64, 41])
    mask = O.lengths_to_mask(lens, T)
    g = torch.Generator().manual_seed(42)
    units = torch.randint(4, 1004, (B, T), generator=g).masked_fill(~mask, 0)
    return feat, lens, mask, units


def gen_vae_train_full(lm):
    """The recipe-sized VAE (dim 768, d_h 96, inner 2048, 138.6 M parameters): loss terms and the gradient of every parameter
    (checksums + 256 strided samples each; whole tensors up to 2048 elements)."""
    from gen_golden_configs import FULL_VAE

    vsd = O.make_vae_state_dict(FULL_VAE, "full")
    vae = ref_vae(lm, FULL_VAE, vsd)
    feat, lens, mask, units = full_batch(FULL_VAE.dim)
    out = dict(lens=lens, units=units)
    torch.manual_seed(502)
    with record_draws() as rec:
        loss, parts, logits = vae_criterion_loss(lm, vae, feat, units, mask, lens)
    out["post_noise"] = rec.draws[0].transpose(1, 2).contiguous()
    loss.backward()
    out["loss"] = loss.detach()
    out.update({k: v.detach() for k, v in parts.items()})
    out["logits_head"] = logits.detach()[:, :4, :64]
    grad_record("g/", [(n, p.grad) for n, p in vae.named_parameters()], out, small=2048, samples=256)
    save("vae_train_full", **out)


def gen_eps_train_full(lm):
    """The recipe-sized eps-predictor (dim 512, depth 12, inner 1365 -> 1408, 57 k-wide conditioning projection, 260.6 M parameters)
    through the frozen recipe-sized VAE: the loss dict and the gradient of every eps-predictor parameter, same storage."""
    from gen_golden_configs import FULL_EPS, FULL_VAE

    esd, vsd = O.make_eps_state_dict(FULL_EPS, "full"), O.make_vae_state_dict(FULL_VAE, "full")
    vae = ref_vae(lm, FULL_VAE, vsd)
    for p in vae.parameters():
        p.requires_grad = False
    ldm = lm.LatentDiscreteModel(types.SimpleNamespace(encoder=vae), FULL_EPS.dim, FULL_VAE.z, timesteps=200, multitask=True)
    ldm.model.load_state_dict(dict(esd, **{"pos_embed._float_tensor": torch.zeros(1)}), strict=True)
    ldm.eval()
    feat, lens, mask, units = full_batch(FULL_VAE.dim)
    out = dict(lens=lens, units=units)
    torch.manual_seed(778)
    with record_draws() as rec:
        ld = ldm(feat, units, tgt_mask=mask)
    assert len(rec.draws) == 4, len(rec.draws)
    out.update(times=rec.draws[0], post_noise=rec.draws[1].transpose(1, 2).contiguous(), jitter=rec.draws[2].contiguous(),
               true_noise=rec.draws[3].contiguous(), **{"loss_" + k: v.detach() for k, v in ld.items()})
    ld["total_loss"].backward()
    grad_record("g/", [(n, p.grad) for n, p in ldm.model.named_parameters() if p.grad is not None], out, small=2048, samples=256)
    out["no_grad_names"] = np.array([n for n, p in ldm.model.named_parameters() if p.grad is None])
    save("eps_train_full", **out)


class inject_draws:
    """While the reference runs, every torch.randn / randn_like / randint call returns a portable seeded tensor (seed base + call
    index) instead of a draw off the global generator; `calls` lists (kind, shape, seed) in call order for the test to regenerate."""

    def __init__(self, base):
        self.base, self.calls = base, []

    def __enter__(self):
        self._orig = {n: getattr(torch, n) for n in ("randn", "randn_like", "randint")}

        def randn(*shape, **k):
            shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else tuple(shape)
            seed = self.base + len(self.calls)
            self.calls.append(("randn", shape, seed))
            return self._orig["randn"](*shape, generator=torch.Generator().manual_seed(seed))  # == gen_golden_configs.seeded(shape, seed)

        def randn_like(x, **k):
            return randn(*x.shape)

        def randint(low, high, size, **k):
            seed = self.base + len(self.calls)
            self.calls.append(("randint", tuple(size), seed))
            return self._orig["randint"](low, high, tuple(size), generator=torch.Generator().manual_seed(seed))

        torch.randn, torch.randn_like, torch.randint = randn, randn_like, randint
        return self

    def __exit__(self, *exc):
        for n, f in self._orig.items():
            setattr(torch, n, f)


def bench_batch(B, T, dim, seed):
    """One batch of bench.py's make_train_batches shape class: lengths U[64, 512] sorted longest first (the first one T), features
    N(0, 1) zero-padded, units U{4..1003} with 0 on the pads."""
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(64, T + 1, (B,), generator=g).sort(descending=True).values
    lens[0] = T
    mask = O.lengths_to_mask(lens, T)
    feat = seeded((B, T, dim), seed + 1) * mask.unsqueeze(-1)
    units = torch.randint(4, 1004, (B, T), generator=g).masked_fill(~mask, 0)
    return feat, lens, mask, units


def gen_vae_train_batch(lm):
    """speech_vae_decoder_loss on the recipe-sized VAE at B = 24, T = 512 (bench.py --max-tokens 15000)."""
    from gen_golden_configs import FULL_VAE

    vae = ref_vae(lm, FULL_VAE, O.make_vae_state_dict(FULL_VAE, "full"))
    feat, lens, mask, units = bench_batch(24, 512, FULL_VAE.dim, 4300)
    out = dict(lens=lens, units=units, batch_seed=np.array(4300))
    with inject_draws(9100) as inj:
        loss, parts, logits = vae_criterion_loss(lm, vae, feat, units, mask, lens)
    assert [c[0] for c in inj.calls] == ["randn"] and inj.calls[0][1] == (24, FULL_VAE.z, 512), inj.calls
    out["post_noise_seed"], out["post_noise_shape"] = np.array(inj.calls[0][2]), np.array(inj.calls[0][1])  # drawn [B, z, T] (distributions.py:38)
    loss.backward()
    out["loss"] = loss.detach()
    out.update({k: v.detach() for k, v in parts.items()})
    out["logits_head"] = logits.detach()[:, :4, :64]
    grad_record("g/", [(n, p.grad) for n, p in vae.named_parameters()], out, small=2048, samples=256)
    save("vae_train_batch", **out)


def gen_eps_train_batch(lm):
    """ddpm_discrete_loss on the recipe-sized eps-predictor through the frozen VAE at B = 16, T = 512 (bench.py --max-tokens 12000)."""
    from gen_golden_configs import FULL_EPS, FULL_VAE

    esd, vsd = O.make_eps_state_dict(FULL_EPS, "full"), O.make_vae_state_dict(FULL_VAE, "full")
    vae = ref_vae(lm, FULL_VAE, vsd)
    for p in vae.parameters():
        p.requires_grad = False
    ldm = lm.LatentDiscreteModel(types.SimpleNamespace(encoder=vae), FULL_EPS.dim, FULL_VAE.z, timesteps=200, multitask=True)
    ldm.model.load_state_dict(dict(esd, **{"pos_embed._float_tensor": torch.zeros(1)}), strict=True)
    ldm.eval()
    feat, lens, mask, units = bench_batch(16, 512, FULL_VAE.dim, 4400)
    out = dict(lens=lens, units=units, batch_seed=np.array(4400))
    with inject_draws(9200) as inj:
        ld = ldm(feat, units, tgt_mask=mask)
    assert [c[0] for c in inj.calls] == ["randint", "randn", "randn", "randn"], inj.calls  # t, posterior noise [B,z,T], jitter, true noise
    out["draw_seeds"] = np.array([c[2] for c in inj.calls])
    out["draw_shapes"] = np.array([list(c[1]) + [0] * (3 - len(c[1])) for c in inj.calls])
    out["times"] = torch.randint(1, 200, (16,), generator=torch.Generator().manual_seed(int(inj.calls[0][2])))  # (the draw itself: 16 integers)
    out.update({"loss_" + k: v.detach() for k, v in ld.items()})
    ld["total_loss"].backward()
    grad_record("g/", [(n, p.grad) for n, p in ldm.model.named_parameters() if p.grad is not None], out, small=2048, samples=256)
    out["no_grad_names"] = np.array([n for n, p in ldm.model.named_parameters() if p.grad is None])
    save("eps_train_batch", **out)


def main():
    torch.set_grad_enabled(True)
    lm, _ = ref_loader.load_reference()
    if "--full-batch" in sys.argv:  # recipe-sized models at the benchmark's batch shapes: ~10 minutes and tens of GB on the CPU
        if "eps" not in sys.argv:
            gen_vae_train_batch(lm)
        if "vae" not in sys.argv:
            gen_eps_train_batch(lm)
        return
    if "--full" in sys.argv:  # recipe-sized models: a few minutes of CPU
        gen_vae_train_full(lm)
        gen_eps_train_full(lm)
        return
    gen_vae_train(lm)
    gen_eps_train(lm)


if __name__ == "__main__":
    main()
