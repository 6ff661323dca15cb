"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (diffnorm_amd/).

Training-side checker: gradients of the CPU oracle's loss functions by torch autograd (the oracle is a functional restatement
in plain torch ops, so autograd differentiates exactly what the reference's modules compute: reference
latent_module.py:1118-1142 + fairseq/criterions/speech_vae_decoder_loss.py:45-95 for the VAE, latent_module.py:1514-1613 for
the diffusion loss), and the comparison of a set of named gradients with the fixtures that oracle/gen_golden_train.py took from
the real reference (tests/golden/{vae_train,eps_train}.npz).  Pinned by tests/test_oracle_train.py."""
from typing import Dict

import numpy as np
import torch

import diffnorm_oracle as O


def probe_vector(n: int) -> torch.Tensor:
    """The fixed +-1 pattern of the fixtures' dot-product checksum (gen_golden_train.probe_vector)."""
    i = torch.arange(n, dtype=torch.int64)
    return (((i * 2654435761 + 12345) >> 7) & 1).to(torch.float64) * 2 - 1


def _leaf_copy(sd):
    return {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}


def vae_loss_and_grads(sd, cfg, feat, units, lens, noise):
    """-> (loss dict of floats, {name: grad}) of the VAE criterion."""
    leaf = _leaf_copy(sd)
    out = O.vae_criterion(leaf, cfg, feat, units, lens, noise)
    out["loss"].backward()
    return {k: float(v.detach()) for k, v in out.items()}, {k: v.grad for k, v in leaf.items() if v.grad is not None}


def eps_loss_and_grads(eps_sd, eps_cfg, vae_sd, vae_cfg, timesteps, feat, units, mask, times, post_noise, jitter, true_noise):
    """-> (loss dict, {name: grad}) of LatentDiscreteModel.forward's total_loss w.r.t. the eps-predictor (VAE frozen)."""
    leaf = _leaf_copy(eps_sd)
    out = O.diffusion_train_forward(leaf, eps_cfg, vae_sd, vae_cfg, timesteps, feat, units, mask, times, post_noise, jitter,
                                    true_noise, multitask=True)
    out["total_loss"].backward()
    return {k: float(v.detach()) for k, v in out.items()}, {k: v.grad for k, v in leaf.items() if v.grad is not None}


def compare_grads(grads: Dict[str, torch.Tensor], golden, prefix: str = "g/", rtol: float = 1e-3):
    """Every tensor named in the fixture against `grads`: whole tensor or strided sample within rtol of the tensor's own
    scale (max |g|), and the three checksums [sum, l2, probe dot] within rtol of the l2 norm.  Returns the worst relative error."""
    names = [str(n) for n in golden[prefix + "names"]]
    worst = 0.0
    for name in names:
        assert name in grads, f"gradient for {name} is missing"
        g = grads[name].detach().double().cpu()
        chk = golden[f"{prefix}chk/{name}"]
        n = g.numel()
        l2 = max(chk[1], 1e-30)
        mine = np.array([g.sum().item(), g.norm().item(), (g.flatten() * probe_vector(n)).sum().item()])
        # sum / probe-dot are sums of n terms of size ~l2/sqrt(n): compare against l2 (their natural scale)
        cerr = np.abs(mine - chk).max() / l2
        if f"{prefix}full/{name}" in golden:
            ref = torch.from_numpy(golden[f"{prefix}full/{name}"]).double()
            got = g.reshape(ref.shape)
        else:
            ref = torch.from_numpy(golden[f"{prefix}samp/{name}"]).double()
            stride = (n + ref.numel() - 1) // ref.numel()
            got = g.flatten()[::stride]
        scale = max(ref.abs().max().item(), 1e-30)
        eerr = (got - ref).abs().max().item() / scale
        worst = max(worst, cerr, eerr)
        assert cerr <= rtol, f"{name}: checksum error {cerr:.3e} (mine {mine}, reference {chk})"
        assert eerr <= rtol, f"{name}: element error {eerr:.3e} relative to max |g| = {scale:.3e}"
    return worst
