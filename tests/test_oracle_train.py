"""Pins the oracle's autograd gradients (oracle/train_oracle.py) to the gradients the REAL reference produced
(oracle/gen_golden_train.py -> tests/golden/{vae_train,eps_train}.npz).  CPU only, no reference needed."""
import numpy as np
import torch

import diffnorm_oracle as O
import train_oracle as TO
from gen_golden_configs import CHAIN_EPS, CHAIN_VAE, seeded


def _batch(g):
    feat = seeded((3, 48, CHAIN_VAE.dim), 31)
    lens = torch.from_numpy(g["lens"])
    return feat, lens, O.lengths_to_mask(lens, 48), torch.from_numpy(g["units"])


def test_vae_criterion_gradients_match_reference(golden):
    g = golden("vae_train")
    feat, lens, mask, units = _batch(g)
    sd = O.make_vae_state_dict(CHAIN_VAE, "train")
    losses, grads = TO.vae_loss_and_grads(sd, CHAIN_VAE, feat, units, lens, torch.from_numpy(g["post_noise"]))
    for k in ("loss", "nll_loss", "mse_loss", "kl_loss", "acc"):
        assert abs(losses[k] - float(g[k])) <= 2e-5 * max(1.0, abs(float(g[k]))), (k, losses[k], float(g[k]))
    assert len(g["g/names"]) == len(sd) == len(grads)
    worst = TO.compare_grads(grads, g, "g/", rtol=2e-4)
    total = float(torch.sqrt(sum(v.double().pow(2).sum() for v in grads.values())))
    assert abs(total - float(g["g/total_norm"])) <= 1e-4 * float(g["g/total_norm"])
    print("worst relative gradient error", worst)


def test_diffusion_loss_gradients_match_reference(golden):
    g = golden("eps_train")
    feat, lens, mask, units = _batch(g)
    esd = O.make_eps_state_dict(CHAIN_EPS, "train")
    vsd = O.make_vae_state_dict(CHAIN_VAE, "train")
    T = lambda k: torch.from_numpy(g[k])
    losses, grads = TO.eps_loss_and_grads(esd, CHAIN_EPS, vsd, CHAIN_VAE, 200, feat, units, mask, T("times"), T("post_noise"),
                                          T("jitter"), T("true_noise"))
    for k in ("total_loss", "nll_loss", "recon_mse_loss", "noise_loss", "acc"):
        assert abs(losses[k] - float(g["loss_" + k])) <= 2e-5 * max(1.0, abs(float(g["loss_" + k]))), k
    assert len(g["no_grad_names"]) == 0
    TO.compare_grads(grads, g, "g/", rtol=5e-4)
