"""CPU: host logic of the train-mode attention dropout (the kernels' parity is tests/test_hip_train*.py): the seed fields the
engines hand to the kernels, the host restatement of the counter-hash mask, and the oracle's train-mode context."""
import types

import torch

import diffnorm_oracle as O
from dropout_mask import dropout_keep_mask, layer_keep


def test_dropout_fields_advance_per_forward_and_stay_off_in_eval():
    from diffnorm_amd.training import _dropout_fields

    eng = types.SimpleNamespace(attn_dropout=0.0, dropout_seed=5)
    assert _dropout_fields(eng) == (0.0, 0, 0, 0) and not hasattr(eng, "_dropout_calls")  # eval: nothing drawn
    eng.attn_dropout = 0.1
    a, b = _dropout_fields(eng), _dropout_fields(eng)
    assert a[0] == b[0] == 0.1 and (a[1], a[2]) != (b[1], b[2])  # every micro-batch its own mask
    again = types.SimpleNamespace(attn_dropout=0.1, dropout_seed=5)
    assert _dropout_fields(again) == a  # a function of (seed, forwards so far) only
    other = types.SimpleNamespace(attn_dropout=0.1, dropout_seed=6)
    assert _dropout_fields(other)[1:3] != a[1:3]
    for f in (a, b):
        assert 0 <= f[1] < 2 ** 32 and 0 <= f[2] < 2 ** 32


def test_mask_restatement_is_bernoulli_and_positional():
    p, seed = 0.1, 0x0123456789ABCDEF
    m = dropout_keep_mask(3, 4, 50, 70, p, seed)
    assert m.shape == (3, 4, 50, 70) and m.dtype == torch.bool
    rate = 1.0 - m.double().mean().item()
    assert abs(rate - p) < 4 * (p * (1 - p) / m.numel()) ** 0.5
    # entry (b, h, i, j) depends on its own indices only: a longer key axis extends the mask, a bigger batch appends to it
    assert torch.equal(dropout_keep_mask(3, 4, 50, 90, p, seed)[..., :70], m)
    assert torch.equal(dropout_keep_mask(5, 4, 50, 70, p, seed)[:3], m)
    assert not torch.equal(dropout_keep_mask(3, 4, 50, 70, p, seed + 1), m)
    assert dropout_keep_mask(2, 2, 8, 8, 0.0, seed).all()
    k0, k1 = layer_keep(p, 7, 9)(0, 2, 2, 16, 16), layer_keep(p, 7, 9)(1, 2, 2, 16, 16)
    assert not torch.equal(k0, k1) and torch.equal(k1, dropout_keep_mask(2, 2, 16, 16, p, (10 << 32) | 7))


def test_oracle_train_mode_context_applies_to_the_named_transformer_only():
    from gen_golden_configs import TINY_EPS

    sd = O.make_eps_state_dict(TINY_EPS, "tiny")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 24, TINY_EPS.latent_dim, generator=g)
    t = torch.tensor([3, 700])
    mask = O.lengths_to_mask(torch.tensor([24, 17]), 24)
    base = O.eps_forward(sd, TINY_EPS, x, t, mask)
    keep_all = lambda layer, B, H, T, Tk: torch.ones(B, H, T, Tk, dtype=torch.bool)
    with O.attention_dropout("eps", 0.0, keep_all):
        assert torch.equal(O.eps_forward(sd, TINY_EPS, x, t, mask), base)
    with O.attention_dropout("vae", 0.5, layer_keep(0.5, 1, 2)):  # another transformer's train mode: no effect here
        assert torch.equal(O.eps_forward(sd, TINY_EPS, x, t, mask), base)
    with O.attention_dropout("eps", 0.5, layer_keep(0.5, 1, 2)):
        dropped = O.eps_forward(sd, TINY_EPS, x, t, mask)
    assert not torch.equal(dropped, base) and torch.isfinite(dropped).all()
    assert torch.equal(O.eps_forward(sd, TINY_EPS, x, t, mask), base)  # the context restores eval mode
