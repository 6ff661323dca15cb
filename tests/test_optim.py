"""Optimizer step (first piece).  CPU: the oracle restatement against the golden vectors the reference's own Adam /
clip_grad_norm_ / InverseSquareRootSchedule produced (oracle/gen_golden_optim.py).  GPU: the HIP kernels against the same
vectors through the C ABI, and at a large size against the same arithmetic in torch fp32."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import optim as O  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "optim.npz"))
STEPS, WARMUP = (int(x) for x in G["meta"])
LR, WARMUP_INIT, B1, B2, EPS, CLIP = (float(x) for x in G["hyper"])
SIZES = [37 * 19, 1001, 8 * 3 * 5]  # the three tensors of the golden run, concatenated


def split(a):
    return np.split(a, np.cumsum(SIZES)[:-1])


@pytest.mark.parametrize("prefix,wd", [("", 0.0), ("wd_", 0.01)])
def test_oracle_matches_reference_golden(prefix, wd):
    p = G[prefix + "p0"]
    m, v = np.zeros_like(p), np.zeros_like(p)
    clipped = 0
    for it in range(STEPS):
        g = G[f"{prefix}g{it}"]
        norm = O.total_norm(split(g))
        assert abs(norm - G[f"{prefix}norm{it}"]) <= 2e-6 * G[f"{prefix}norm{it}"]
        lr = O.inverse_sqrt_lr(it, LR, WARMUP, WARMUP_INIT)
        assert lr == pytest.approx(float(G[f"{prefix}lr{it}"]), rel=1e-12)
        c = O.clip_coef(norm, CLIP)
        clipped += c < 1
        p, m, v = O.adam_step(p, g * c, m, v, it + 1, lr, (B1, B2), EPS, wd)
        for got, want in ((p, G[f"{prefix}p{it + 1}"]), (m, G[f"{prefix}m{it + 1}"]), (v, G[f"{prefix}v{it + 1}"])):
            assert np.max(np.abs(got - want)) <= 2e-6 * max(1e-3, np.max(np.abs(want)))
    assert 0 < clipped < STEPS  # the golden run has both clipped and unclipped updates


@pytest.mark.gpu
@pytest.mark.parametrize("prefix,wd", [("", 0.0), ("wd_", 0.01)])
def test_hip_adam_matches_reference_golden(prefix, wd):
    from diffnorm_amd import optim

    dev = "cuda:0"
    p = torch.from_numpy(G[prefix + "p0"].copy()).to(dev)
    shadow = torch.zeros(p.numel(), device=dev, dtype=torch.bfloat16)
    opt = optim.Adam(p, lr=LR, betas=(B1, B2), eps=EPS, weight_decay=wd, clip_norm=CLIP, bf16_copy=shadow)
    sched = optim.InverseSquareRootSchedule(LR, WARMUP, WARMUP_INIT)
    for it in range(STEPS):
        g = torch.from_numpy(G[f"{prefix}g{it}"].copy()).to(dev)
        opt.set_lr(sched.step_update(it))
        assert opt.get_lr() == pytest.approx(float(G[f"{prefix}lr{it}"]), rel=1e-12)
        norm = opt.step(g).item()
        assert abs(norm - G[f"{prefix}norm{it}"]) <= 2e-6 * G[f"{prefix}norm{it}"]
        for got, want in ((p, G[f"{prefix}p{it + 1}"]), (opt.exp_avg, G[f"{prefix}m{it + 1}"]), (opt.exp_avg_sq, G[f"{prefix}v{it + 1}"])):
            assert np.max(np.abs(got.cpu().numpy() - want)) <= 2e-6 * max(1e-3, np.max(np.abs(want)))  # fp32 tolerance
        assert torch.equal(shadow, p.to(torch.bfloat16))  # the bf16 working copy is the rounded fp32 master
        assert torch.equal(g.cpu(), torch.from_numpy(G[f"{prefix}g{it}"]))  # the gradient buffer is left unscaled


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 3, 4, 1027, 50_000_003])
def test_hip_adam_sizes_and_norm_accumulation(n):
    """Ragged and large sizes against the same arithmetic in torch fp32 on the GPU; the norm over two buffers accumulates."""
    from diffnorm_amd import optim

    dev = "cuda:0"
    gen = torch.Generator(device=dev).manual_seed(n)
    p = torch.randn(n, device=dev, generator=gen)
    g = torch.randn(n, device=dev, generator=gen) * 3e-3
    opt = optim.Adam(p.clone(), lr=3e-4, betas=(0.9, 0.98), eps=1e-8, clip_norm=2.0)
    rp, rm, rv = p.clone(), torch.zeros_like(p), torch.zeros_like(p)
    for step in (1, 2):
        norm = opt.step(g)
        tn = torch.norm(g, p=2, dtype=torch.float32)
        assert abs(norm.item() - tn.item()) <= 1e-5 * tn.item()
        gc = g * (2.0 / (tn + 1e-6)).clamp_(max=1)
        rm.mul_(0.9).add_(gc, alpha=1 - 0.9)
        rv.mul_(0.98).addcmul_(gc, gc, value=1 - 0.98)
        step_size = 3e-4 * (1 - 0.98 ** step) ** 0.5 / (1 - 0.9 ** step)
        rp.addcdiv_(rm, rv.sqrt().add_(1e-8), value=-step_size)
        assert (opt.params - rp).abs().max().item() <= 1e-6 * max(1.0, rp.abs().max().item())
        assert (opt.exp_avg_sq - rv).abs().max().item() <= 1e-6 * rv.abs().max().item() + 1e-12
    a = opt.grad_sumsq(g).clone()
    b = opt.grad_sumsq(g, accumulate=True)
    assert b.item() == pytest.approx(2 * a.item(), rel=1e-6)
    assert torch.equal(opt.grad_sumsq(g), a)  # fixed summation order: bit-reproducible


@pytest.mark.gpu
def test_hip_adam_rejects_bad_arguments():
    from diffnorm_amd import _lib, optim

    opt = optim.Adam(torch.zeros(8, device="cuda:0"))
    with pytest.raises(ValueError):
        opt.step(torch.zeros(9, device="cuda:0"))
    with pytest.raises(ValueError):
        optim.Adam(torch.zeros(8, device="cuda:0", dtype=torch.float64))
    opt.betas = (1.5, 0.9)
    with pytest.raises(_lib.DiffNormHipError):
        opt.step(torch.zeros(8, device="cuda:0"))
