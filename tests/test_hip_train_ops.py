"""GPU parity of the backward ops (through the C ABI) against torch autograd of the CPU oracle's functions on the
same seeded inputs.  f32 mode: exact-fp32 MFMA / fp32 pointwise -> 1e-4 relative; bf16 mode: operands rounded to bf16 on both
sides, the remaining difference is the bf16 rounding of P / dS inside the attention kernels and of stored outputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import diffnorm_oracle as O
from dropout_mask import dropout_keep_mask

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def seeded(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def bf16r(t):
    return t.to(torch.bfloat16).float()


def tdt(dtype):
    return torch.bfloat16 if dtype == "bf16" else torch.float32


def act(t, dtype):
    return t.to(DEV, tdt(dtype)).contiguous()


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


@pytest.fixture(scope="module")
def ops():
    from diffnorm_amd import _lib, ops, packing

    _lib.load()
    return ops, packing, _lib


def ref_attention(q, k, v, lens, heads):
    """Attend.forward non-flash (latent_module.py:299-343) in float64, [B,T,h*d] layout."""
    B, T, hd = q.shape
    d = hd // heads
    split = lambda t: t.view(B, T, heads, d).transpose(1, 2)
    sim = torch.einsum("bhid,bhjd->bhij", split(q), split(k)) * d ** -0.5
    mask = O.lengths_to_mask(lens, T)
    sim = sim.masked_fill(~mask.view(B, 1, 1, T), -torch.finfo(sim.dtype).max)
    out = torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), split(v))
    return out.transpose(1, 2).reshape(B, T, hd)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("heads,dh,B,T,lens", [(4, 16, 3, 40, [40, 23, 1]), (2, 64, 2, 150, [150, 77]), (8, 96, 2, 200, [130, 200]),
                                               (2, 64, 1, 300, [257]), (2, 32, 2, 64, [64, 0])])
def test_attention_backward(ops, dtype, heads, dh, B, T, lens):
    ops_, packing, _lib = ops
    hd = heads * dh
    rnd = bf16r if dtype == "bf16" else (lambda t: t)
    q, k, v, do = (rnd(seeded((B, T, hd), 50 + i)) for i in range(4))
    lens_t = torch.tensor(lens)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    want_o = ref_attention(qd, kd, vd, lens_t, heads)
    want_o.backward(do.double())
    qkv = act(torch.cat([q, k, v], dim=-1).view(B * T, 3 * hd), dtype)
    out = torch.empty(B * T, hd, dtype=qkv.dtype, device=DEV)
    l32 = lens_t.to(DEV, torch.int32)
    _, lse = ops_.attention_fwd_lse(qkv, qkv[:, hd:], qkv[:, 2 * hd:], out, B, T, heads, dh, l32, ldq=3 * hd, ldk=3 * hd, ldv=3 * hd)
    tol_f = 2e-5 if dtype == "f32" else 1e-2
    assert relerr(out.float().view(B, T, hd), want_o) < tol_f
    # lse: log2-domain log-sum-exp of the scaled, key-masked scores
    if min(lens) > 0:
        sim = torch.einsum("bihd,bjhd->bhij", qd.detach().view(B, T, heads, dh), kd.detach().view(B, T, heads, dh)) * dh ** -0.5
        mask = O.lengths_to_mask(lens_t, T)
        want_lse = torch.logsumexp(sim.masked_fill(~mask.view(B, 1, 1, T), -1e300), dim=-1) / np.log(2.0)
        assert (lse.cpu().double() - want_lse).abs().max().item() < (1e-4 if dtype == "f32" else 3e-2)
    doa = act(do.view(B * T, hd), dtype)
    dq, dk, dv = ops_.attention_backward(qkv, qkv[:, hd:], qkv[:, 2 * hd:], out, doa, lse, B, T, heads, dh, l32, ld_qkv=3 * hd)
    tol = 1e-4 if dtype == "f32" else 2e-2
    for name, got, want in (("dq", dq, qd.grad), ("dk", dk, kd.grad), ("dv", dv, vd.grad)):
        e = relerr(got.float().reshape(B, T, hd), want)
        assert e < tol, (name, e)
    # bit-reproducible (no atomics)
    dq2, dk2, dv2 = ops_.attention_backward(qkv, qkv[:, hd:], qkv[:, 2 * hd:], out, doa, lse, B, T, heads, dh, l32, ld_qkv=3 * hd)
    assert torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("heads,dh,B,T,lens,p", [(4, 16, 3, 40, [40, 23, 1], 0.1), (2, 64, 2, 150, [150, 77], 0.1), (8, 64, 2, 200, [130, 200], 0.1),
                                                 (2, 64, 1, 300, [257], 0.5), (2, 32, 2, 64, [64, 0], 0.25)])
def test_attention_dropout_forward_backward(ops, dtype, heads, dh, B, T, lens, p):
    """Train mode of Attend (latent_module.py:338: attn = dropout(softmax(sim)); out = attn @ v): the kernels' counter-hash mask,
    restated on the host, put into the float64 reference -- forward and the three gradients agree, the drop rate is p, and the
    backward re-derives the very mask the forward used."""
    ops_, packing, _lib = ops
    hd = heads * dh
    seed = 0x1234ABCD5678EF01 + T
    rnd = bf16r if dtype == "bf16" else (lambda t: t)
    q, k, v, do = (rnd(seeded((B, T, hd), 60 + i)) for i in range(4))
    lens_t = torch.tensor(lens)
    keep = dropout_keep_mask(B, heads, T, T, p, seed)
    valid = O.lengths_to_mask(lens_t, T).view(B, 1, 1, T).expand(B, heads, T, T)
    rate = 1.0 - keep[valid].double().mean().item()
    assert abs(rate - p) < 4 * (p * (1 - p) / max(int(valid.sum()), 1)) ** 0.5 + 1e-3, rate
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    split = lambda t: t.view(B, T, heads, dh).transpose(1, 2)
    sim = torch.einsum("bhid,bhjd->bhij", split(qd), split(kd)) * dh ** -0.5
    sim = sim.masked_fill(~valid, -torch.finfo(sim.dtype).max)
    attn = sim.softmax(dim=-1) * keep.double() / (1.0 - p)
    want_o = torch.einsum("bhij,bhjd->bhid", attn, split(vd)).transpose(1, 2).reshape(B, T, hd)
    want_o.backward(do.double())
    qkv = act(torch.cat([q, k, v], dim=-1).view(B * T, 3 * hd), dtype)
    out = torch.empty(B * T, hd, dtype=qkv.dtype, device=DEV)
    l32 = lens_t.to(DEV, torch.int32)
    args = (B, T, heads, dh, l32)
    _, lse = ops_.attention_fwd_lse(qkv, qkv[:, hd:], qkv[:, 2 * hd:], out, *args, ldq=3 * hd, ldk=3 * hd, ldv=3 * hd, dropout_p=p, seed=seed)
    live = torch.tensor([n > 0 for n in lens]).view(B, 1, 1)
    sel = (live & torch.ones(B, T, 1, dtype=torch.bool)).expand(B, T, hd)  # an empty sequence has no defined softmax: real ones only
    tol_f = 2e-5 if dtype == "f32" else 1.5e-2
    assert relerr(out.float().view(B, T, hd).cpu()[sel], want_o.detach()[sel]) < tol_f
    # the log-sum-exp is that of the undropped softmax: same as without dropout
    out0 = torch.empty_like(out)
    _, lse0 = ops_.attention_fwd_lse(qkv, qkv[:, hd:], qkv[:, 2 * hd:], out0, *args, ldq=3 * hd, ldk=3 * hd, ldv=3 * hd)
    assert torch.equal(lse, lse0) and not torch.equal(out, out0)
    doa = act(do.view(B * T, hd), dtype)
    dq, dk, dv = ops_.attention_backward(qkv, qkv[:, hd:], qkv[:, 2 * hd:], out, doa, lse, *args, ld_qkv=3 * hd, dropout_p=p, seed=seed)
    tol = 1e-4 if dtype == "f32" else 2.5e-2
    for name, got, want in (("dq", dq, qd.grad), ("dk", dk, kd.grad), ("dv", dv, vd.grad)):
        e = relerr(got.float().reshape(B, T, hd).cpu()[sel], want[sel])
        assert e < tol, (name, e)
    # another seed is another mask
    out2 = torch.empty_like(out)
    ops_.attention_fwd_lse(qkv, qkv[:, hd:], qkv[:, 2 * hd:], out2, *args, ldq=3 * hd, ldk=3 * hd, ldv=3 * hd, dropout_p=p, seed=seed + (1 << 32))
    assert not torch.equal(out, out2)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("mode", ["learned", "adaptive", "plain"])
@pytest.mark.parametrize("B,T,D", [(3, 37, 64), (2, 300, 768), (5, 9, 192)])
def test_rmsnorm_backward(ops, dtype, mode, B, T, D):
    ops_, packing, _lib = ops
    Dp = packing.padk(D)
    x = seeded((B, T, D), 1).double().requires_grad_(True)
    dy = (bf16r if dtype == "bf16" else (lambda t: t))(seeded((B, T, D), 2))
    dres = seeded((B, T, D), 3)
    gamma = (1 + 0.3 * seeded((D,), 4)).double().requires_grad_(True) if mode == "learned" else None
    gb = seeded((B, 2 * Dp), 5).double().requires_grad_(True) if mode == "adaptive" else None
    y = F.normalize(x, dim=-1) * D ** 0.5
    if gamma is not None:
        y = y * gamma
    if gb is not None:
        y = y * gb[:, None, :D] + gb[:, None, Dp:Dp + D]
    y.backward(dy.double())
    xa = torch.zeros(B * T, Dp)
    xa[:, :D] = x.detach().float().view(B * T, D)
    dya = torch.zeros(B * T, Dp)
    dya[:, :D] = dy.view(B * T, D)
    dra = torch.zeros(B * T, Dp)
    dra[:, :D] = dres.view(B * T, D)
    dgamma = torch.full((D,), 0.5, device=DEV) if mode == "learned" else None  # accumulates onto what is there
    dgb = torch.zeros(B, 2 * Dp, device=DEV) if mode == "adaptive" else None
    dx, dx_act = ops_.rmsnorm_backward(xa.to(DEV), act(dya, dtype), B, T, D, gamma=gamma.detach().float().to(DEV) if gamma is not None else None,
                                       gamma_beta=gb.detach().float().to(DEV) if gb is not None else None, gb_half=Dp, dres=dra.to(DEV),
                                       act_dtype=tdt(dtype), dgamma=dgamma, dgamma_beta=dgb)
    want_dx = x.grad.view(B * T, D) + dres.view(B * T, D).double()
    assert relerr(dx[:, :D], want_dx) < 2e-5
    assert dx[:, D:].abs().max().item() == 0 if Dp > D else True
    assert relerr(dx_act[:, :D].float(), want_dx) < (2e-5 if dtype == "f32" else 1e-2)
    if mode == "learned":
        assert relerr(dgamma - 0.5, gamma.grad) < 1e-4
    if mode == "adaptive":
        assert relerr(dgb[:, :D], gb.grad[:, :D]) < 1e-4
        assert relerr(dgb[:, Dp:Dp + D], gb.grad[:, Dp:Dp + D]) < 1e-4


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("film", [False, True])
def test_gate_forward_backward(ops, dtype, film):
    ops_, packing, _lib = ops
    B, T, Cp = 3, 50, 128
    rnd = bf16r if dtype == "bf16" else (lambda t: t)
    h = rnd(seeded((B, T, Cp), 1, 2.0)).double().requires_grad_(True)
    res = rnd(seeded((B, T, Cp), 2))
    dout = rnd(seeded((B, T, Cp), 3))
    gb = seeded((B, 2 * Cp), 4).double().requires_grad_(True) if film else None
    hh = h * gb[:, None, :Cp] + gb[:, None, Cp:] if film else h
    out = torch.tanh(hh) * torch.sigmoid(hh) + res.double()
    out.backward(dout.double())
    gbd = gb.detach().float().to(DEV) if film else None
    got = ops_.gate_forward(act(h.detach().float().view(B * T, Cp), dtype), act(res.view(B * T, Cp), dtype), T, gbd, Cp)
    tol = 1e-5 if dtype == "f32" else 1e-2
    assert relerr(got.float(), out.detach().view(B * T, Cp)) < tol
    r = ops_.gate_backward(act(dout.view(B * T, Cp), dtype), act(h.detach().float().view(B * T, Cp), dtype), T, gbd, Cp, want_rows=film)
    dh, rows = r if film else (r, None)
    assert relerr(dh.float(), h.grad.view(B * T, Cp)) < tol
    if film:
        dgb = ops_.colsum(rows, B, T, 2 * Cp)
        assert relerr(dgb, gb.grad) < 1e-4


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_geglu_forward_backward(ops, dtype):
    ops_, packing, _lib = ops
    M, inner = 70, 100
    ip = packing.padk(inner)
    rnd = bf16r if dtype == "bf16" else (lambda t: t)
    pre = rnd(seeded((M, 2 * inner), 1, 1.5)).double().requires_grad_(True)  # reference column order: [value | gate]
    dout = rnd(seeded((M, inner), 2))
    val, gate = pre.chunk(2, dim=-1)
    out = F.gelu(gate) * val
    out.backward(dout.double())
    rows = packing._geglu_rows(inner)  # packed column -> reference column (or -1)
    keep = rows >= 0
    packed = torch.zeros(M, 2 * ip)
    packed[:, keep] = pre.detach().float()[:, rows[keep]]
    got = ops_.geglu_forward(act(packed, dtype), ip)
    tol = 2e-6 if dtype == "f32" else 1e-2
    assert relerr(got[:, :inner].float(), out.detach()) < tol
    assert got[:, inner:].abs().max().item() == 0
    dpad = torch.zeros(M, ip)
    dpad[:, :inner] = dout
    dpre = ops_.geglu_backward(act(dpad, dtype), act(packed, dtype), ip).float().cpu()
    want = torch.zeros(M, 2 * ip, dtype=torch.double)
    want[:, keep] = pre.grad[:, rows[keep]]
    assert relerr(dpre, want) < (5e-6 if dtype == "f32" else 1e-2)


def test_posterior_backward(ops):
    ops_, packing, _lib = ops
    B, T, Z = 3, 20, 8
    params = seeded((B, T, 2 * Z), 1, 2.0)
    params[0, 0, Z] = 25.0   # outside the clamp: no gradient
    params[0, 1, Z + 1] = -31.0
    noise, dz = seeded((B, T, Z), 2), seeded((B, T, Z), 3)
    lens = torch.tensor([20, 7, 13])
    p = params.double().requires_grad_(True)
    z = O.posterior_sample(p, noise.double())
    kl = O.posterior_kl(p, O.lengths_to_mask(lens, T)).mean()
    ((z * dz.double()).sum() + 1e-2 * kl).backward()
    got = ops_.posterior_backward(params.view(B * T, 2 * Z).to(DEV), noise.view(B * T, Z).to(DEV), dz.view(B * T, Z).to(DEV), Z, T,
                                  lens.to(DEV, torch.int32), 1e-2 / (B * Z * T), torch.float32, 64)
    assert relerr(got[:, :2 * Z], p.grad.view(B * T, 2 * Z)) < 1e-5
    assert got[:, 2 * Z:].abs().max().item() == 0


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_lsce_loss_and_gradient(ops, dtype):
    ops_, packing, _lib = ops
    M, V = 37, 1004
    logits = seeded((M, V), 1, 2.0)
    g = torch.Generator().manual_seed(2)
    tgt = torch.randint(4, V, (M,), generator=g)
    tgt[5] = tgt[11] = 0  # pads
    tgt[3] = int(logits[3].argmax())
    lg = logits.double().requires_grad_(True)
    lprobs = F.log_softmax(lg, dim=-1)
    loss, nll = O.label_smoothed_nll_loss(lprobs, tgt, 0.1, 0)
    gscale = 0.1 / 30
    (loss * gscale).backward()
    rows, dl = ops_.lsce_loss_grad(logits.to(DEV), tgt.to(DEV, torch.int32), 0.1, gscale, act_dtype=tdt(dtype), ldd=1024)
    rows = rows.cpu().double()
    eps_i = 0.1 / (V - 1)
    assert abs(rows[:, 0].sum().item() - nll.item()) < 1e-4 * abs(nll.item())
    assert abs(((1 - 0.1 - eps_i) * rows[:, 0].sum() + eps_i * rows[:, 1].sum()).item() - loss.item()) < 1e-4 * abs(loss.item())
    keep = tgt.ne(0)
    assert rows[:, 3].sum().item() == keep.sum().item()
    assert rows[:, 2].sum().item() == (lprobs.argmax(1)[keep] == tgt[keep]).sum().item() >= 1
    assert relerr(dl[:, :V].float(), lg.grad) < (1e-5 if dtype == "f32" else 1e-2)
    assert dl[:, V:].abs().max().item() == 0


def test_masked_mse_grad(ops):
    ops_, packing, _lib = ops
    B, T, D, Dp = 3, 11, 48, 64
    pred = torch.zeros(B * T, Dp)
    pred[:, :D] = seeded((B * T, D), 1)
    tgt = seeded((B * T, D), 2)
    lens = torch.tensor([11, 4, 9])
    mask = O.lengths_to_mask(lens, T).view(-1)
    p = pred[:, :D].double().requires_grad_(True)
    sel = mask.unsqueeze(1).expand(-1, D)
    mse = F.mse_loss(p[sel], tgt.double()[sel])
    (10 * mse).backward()
    n_valid = int(lens.sum())
    base = seeded((B * T, Dp), 3)
    base[:, D:] = 0
    dpred = base.clone().to(DEV)
    sq, dact = ops_.masked_mse_grad(pred.to(DEV), tgt.to(DEV), T, lens.to(DEV, torch.int32), 10 * 2.0 / (n_valid * D), D, dpred=dpred,
                                    accumulate=True, act_dtype=torch.bfloat16, ld_act=Dp)
    assert abs(sq.sum().item() / (n_valid * D) - mse.item()) < 1e-5 * mse.item()
    assert relerr(dpred[:, :D].cpu() - base[:, :D], p.grad) < 1e-5
    assert relerr(dact.float()[:, :D].cpu(), (p.grad + base[:, :D].double())) < 1e-2
    assert dpred[:, D:].abs().max().item() == 0


def test_reductions_and_transposes(ops):
    ops_, packing, _lib = ops
    src = seeded((6 * 300, 200), 1)
    want = src.view(6, 300, 200).double().sum(1)
    for dt in (torch.float32, torch.bfloat16):
        s = src.to(dt)
        got = ops_.colsum(s.to(DEV), 6, 300, 200)
        assert relerr(got, s.double().view(6, 300, 200).sum(1)) < 1e-5
    acc = torch.ones(6, 200, device=DEV)
    ops_.colsum(src.to(DEV), 6, 300, 200, out=acc, scale=0.5, accumulate=True)
    assert relerr(acc, 1 + 0.5 * want) < 1e-5
    grp = seeded((5, 64, 128), 2)
    assert relerr(ops_.sum_groups(grp.to(DEV)), grp.double().sum(0)) < 1e-6
    for dt in (torch.float32, torch.bfloat16):
        w = seeded((3, 128, 192), 3).to(dt)
        got = ops_.transpose_weights(w.to(DEV), 128, 256).cpu()
        assert got.shape == (3, 256, 128)
        assert torch.equal(got[:, :192, :], w.transpose(1, 2))
        assert got[:, 192:].abs().max().item() == 0


@pytest.mark.parametrize("cin,cout,k,dil,B,T", [(96, 64, 3, 2, 3, 50), (64, 200, 1, 1, 2, 77)])
def test_weight_gradient_f32(ops, cin, cout, k, dil, B, T):
    """The exact-fp32 twin of the weight-gradient path (dn_transpose_pad_f32 + split-K dn_conv_gemm + dn_wgrad_reduce)."""
    ops_, packing, _lib = ops
    x = seeded((B, T, cin), 41)
    w = seeded((cout, cin, k), 42, (1.0 / (cin * k)) ** 0.5).requires_grad_(True)
    dy = seeded((B, T, cout), 43)
    O.causal_conv1d(x, w, None, dil).backward(dy)
    want = w.grad.permute(2, 0, 1)
    pad = lambda t, n: torch.cat([t, torch.zeros(*t.shape[:-1], n - t.shape[-1])], dim=-1)
    xa = pad(x, packing.padk(cin)).view(B * T, -1).to(DEV)
    dya = pad(dy, packing.padk(cout)).view(B * T, -1).to(DEV)
    got = ops_.conv_weight_grad(xa, dya, T, cin, cout, [(k - 1 - j) * dil for j in range(k)])
    assert relerr(got, want) < 1e-5


@pytest.mark.parametrize("B,N,C", [(16, 57344, 2048), (3, 1000, 256), (20, 9000, 516), (40, 700, 1028)])
def test_rows_times_weight(ops, B, N, C):
    """dn_rows_times_weight: out = X [B, N] . W [N, C] with W row-major and streamed once (the data gradient of the eps-predictor's
    conditioning projection, autograd of nn.Linear latent_module.py:841-852): the recipe's shape, ragged row slices, column counts off the
    1024-column workgroup, and more than 32 rows (one stream per 32)."""
    import ctypes as C_

    ops_, _, _lib = ops
    g = torch.Generator().manual_seed(5)
    X = torch.randn(B, N, generator=g)
    W = torch.randn(N, C, generator=g) * N ** -0.5
    Xd, Wd = X.to(DEV), W.to(DEV)
    out = torch.full((B, C), float("nan"), device=DEV)
    lib = _lib.load()
    scratch = torch.empty(int(lib.dn_rows_times_weight_scratch_bytes(B, N, C)) // 4 + 4, device=DEV)
    _lib.check(lib.dn_rows_times_weight(Xd.data_ptr(), N, B, Wd.data_ptr(), C, N, C, out.data_ptr(), scratch.data_ptr(), ops_._stream()),
               "dn_rows_times_weight")
    want = X.double() @ W.double()
    assert (out.cpu().double() - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("cin,cout,k,B,T", [(2048, 2048, 3, 2, 320), (1365, 1365, 3, 3, 100), (704, 2048, 1, 1, 515)])
def test_weight_gradient_on_192_column_tiles_is_bit_identical(ops, hip_option, cin, cout, k, B, T):
    """Option wgrad_k192: the row-major weight-gradient kernel on tiles of 192 k-columns (eight waves of 48 x 128) where that fills the
    chip better than 256 -- the VAE's FFN conv shape 6144 x 2048 (192 -> 256 workgroups), a width off both grids, and a shape where the
    score keeps 256.  Every output sums its frames in the same order: bit-identical to the 256-column tiles, and right."""
    ops_, _, _lib = ops
    M = B * T
    cinp, coutp = (cin + 63) // 64 * 64, (cout + 63) // 64 * 64
    g = torch.Generator().manual_seed(11)
    x = torch.zeros(M, cinp)
    dy = torch.zeros(M, coutp)
    x[:, :cin] = torch.randn(M, cin, generator=g)
    dy[:, :cout] = torch.randn(M, cout, generator=g) * 0.1
    xb, dyb = x.to(DEV, torch.bfloat16), dy.to(DEV, torch.bfloat16)
    shifts = [k - 1 - j for j in range(k)]
    outs = []
    for on in (0, 1):
        hip_option("wgrad_k192", on)
        outs.append(ops_.conv_weight_grad_tn(xb, dyb, T, cin, cout, shifts).cpu())
    assert torch.equal(outs[0], outs[1])
    xr, dr = xb.float().cpu().view(B, T, cinp)[..., :cin], dyb.float().cpu().view(B, T, coutp)[..., :cout]
    for j, sft in enumerate(shifts):
        xs = torch.zeros_like(xr)
        xs[:, sft:] = xr[:, :T - sft]
        want = torch.einsum("btn,btc->nc", dr, xs)
        assert (outs[1][j] - want).abs().max().item() < 2e-3 * max(1.0, want.abs().max().item())

