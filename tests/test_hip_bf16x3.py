"""GPU parity of the DN_BF16X3 arithmetic mode: contraction operands stored as (hi, lo) bf16 pairs in "split rows"
(include/diffnorm_hip.h), three bf16 MFMAs per product into one fp32 accumulator.  The mode exists to meet north_star's
fp32-column budget (1e-3) at bf16-MFMA speed, so every check here is against the FP32 oracle at fp32-class tolerances;
the engine-level golden tests (tests/test_hip_engine.py, test_hip_fullsize.py, test_hip_mirror.py) run it at the same
1e-3 as the exact-fp32 mode.
"""
import numpy as np
import pytest
import torch

import diffnorm_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def seeded(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def padk(c):
    return (c + 63) // 64 * 64


def pad_cols(t, n):
    out = torch.zeros(*t.shape[:-1], n, dtype=t.dtype)
    out[..., : t.shape[-1]] = t
    return out


def maxerr(a, b):
    return (a.double() - b.double()).abs().max().item()


@pytest.fixture(scope="module")
def ops():
    from diffnorm_amd import _lib, ops, packing

    _lib.load()
    return ops, packing, _lib


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 8])
@pytest.mark.parametrize("cin,cout,k,dil,B,T", [(64, 64, 3, 1, 2, 40), (96, 200, 3, 4, 3, 37), (128, 64, 1, 1, 1, 300),
                                                (64, 128, 3, 64, 2, 50), (192, 704, 3, 1, 3, 100), (1408, 1408, 3, 1, 2, 512)])
def test_causal_conv_gemm_x3(ops, tile, cin, cout, k, dil, B, T):
    """CausalConv1d through split operands, every tile that takes them, fp32 and split outputs, ragged M, K/N padding."""
    ops_, packing, _lib = ops
    x = seeded((B, T, cin), 1)
    w = seeded((cout, cin, k), 2, (1.0 / (cin * k)) ** 0.5)
    b = seeded((cout,), 3, 0.1)
    xa = packing.split_rows(pad_cols(x, padk(cin)).view(B * T, -1)).to(DEV)
    W = packing._conv(w, _lib.DN_BF16X3).to(DEV)
    assert W.dtype == torch.bfloat16 and W.shape[-1] == 2 * padk(cin)
    bias = packing._vec(b, padk(cout)).to(DEV)
    terms = [(xa, W[j], (k - 1 - j) * dil) for j in range(k)]
    want = O.causal_conv1d(x, w, b, dil)
    out = torch.full((B * T, padk(cout)), float("nan"), device=DEV)
    ops_.conv_gemm(terms, out, T, padk(cout), bias=bias, x3=True, tile=tile)
    got = out.cpu().view(B, T, -1)
    if padk(cout) > cout:
        assert got[..., cout:].abs().max().item() == 0.0
    err = maxerr(got[..., :cout], want)
    assert err < 1e-4, err  # 2^-16 operands on O(1) sums: measured ~2e-5 (exact-fp32 mode ~1e-6, plain bf16 ~1e-2)
    # the same contraction writing split rows (the form the next contraction stages): hi + lo reproduces the fp32 output to 2^-16
    outs = torch.zeros((B * T, 2 * padk(cout)), dtype=torch.bfloat16, device=DEV)
    ops_.conv_gemm(terms, outs, T, padk(cout), bias=bias, x3=True, tile=tile)
    back = packing.unsplit_rows(outs.cpu())
    assert maxerr(back, out.cpu()) <= out.abs().max().item() * 2.0 ** -15


def test_x3_chain_of_two_contractions_and_epilogues(ops):
    """Linear -> GEGLU (split output) -> causal conv (split in, fp32 out) and a RESADD producer with a split norm output:
    the split rows one contraction writes are the ones the next one stages."""
    ops_, packing, _lib = ops
    B, T, D, inner = 2, 70, 64, 85
    ip = padk(inner)
    x = seeded((B, T, D), 5)
    w_in, b_in = seeded((2 * inner, D), 6, D ** -0.5), seeded((2 * inner,), 7, 0.1)
    w_c, b_c = seeded((inner, inner, 3), 8, (3 * inner) ** -0.5), seeded((inner,), 9, 0.1)
    rows = packing._geglu_rows(inner)
    keep = rows >= 0
    wp, bp = torch.zeros(2 * ip, padk(D)), torch.zeros(2 * ip)
    wp[keep, :D] = w_in[rows[keep]]
    bp[keep] = b_in[rows[keep]]
    xa = packing.split_rows(pad_cols(x, padk(D)).view(B * T, -1)).to(DEV)
    gg = torch.zeros((B * T, 2 * ip), dtype=torch.bfloat16, device=DEV)
    ops_.conv_gemm([(xa, packing.split_rows(wp, weight=True).to(DEV), 0)], gg, T, ip, bias=bp.to(DEV), epilogue=_lib.EPI_GEGLU, x3=True)
    h = torch.nn.functional.linear(x, w_in, b_in)
    want_g = torch.nn.functional.gelu(h[..., inner:]) * h[..., :inner]
    assert maxerr(packing.unsplit_rows(gg.cpu()).view(B, T, -1)[..., :inner], want_g) < 2e-4
    Wc = packing._conv(w_c, _lib.DN_BF16X3).to(DEV)
    out = torch.empty((B * T, ip), device=DEV)
    ops_.conv_gemm([(gg, Wc[j], 2 - j) for j in range(3)], out, T, ip, bias=packing._vec(b_c, ip).to(DEV), x3=True)
    want = O.causal_conv1d(want_g, w_c, b_c, 1)
    assert maxerr(out.cpu().view(B, T, -1)[..., :inner], want) < 2e-4
    # residual-closing contraction + split-norm producer: fp32 stream out, row * gamma as split rows, sums of squares
    w_o = seeded((D, inner), 10, inner ** -0.5)
    res = seeded((B * T, padk(D)), 11)
    gamma = seeded((D,), 12, 0.3) + 1.0
    stream = res.clone().to(DEV)
    xn = torch.zeros((B * T, 2 * padk(D)), dtype=torch.bfloat16, device=DEV)
    ssq = torch.zeros((B * T, 8), device=DEV)
    fc = packing.split_rows(out.cpu()).to(DEV)
    ops_.conv_gemm([(fc, packing._mat(w_o, _lib.DN_BF16X3).to(DEV), 0)], stream, T, padk(D), epilogue=_lib.EPI_RESADD, res=stream,
                   norm_out=xn, norm_D=D, norm_gamma=pad_cols(gamma, padk(D)).to(DEV), norm_ssq=ssq, x3=True)
    want_s = res[:, :D] + want.reshape(B * T, -1) @ w_o.t()
    assert maxerr(stream.cpu()[:, :D], want_s) < 3e-4
    assert maxerr(packing.unsplit_rows(xn.cpu())[:, :D], want_s * gamma) < 5e-4
    assert maxerr(ssq.cpu()[:, : padk(D) // 64].sum(-1), want_s.pow(2).sum(-1)) < 1e-2
