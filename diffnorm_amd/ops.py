"""Op-level Python wrappers over the C ABI (one call = one HIP kernel launch on the current stream).

Used by the host-side mirror of the reference modules and by the parity tests; tensors are torch
CUDA tensors used as plain device buffers.
"""
import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import DN_BF16, DN_F32


def _code(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return DN_BF16
    if t.dtype == torch.float32:
        return DN_F32
    if t.dtype == torch.float16:
        return _lib.DN_F16
    raise TypeError(f"unsupported tensor dtype {t.dtype}")


def _stream():
    return _lib.current_stream()


def conv_gemm(terms: Sequence[Tuple[torch.Tensor, torch.Tensor, int]], out: torch.Tensor, T: int, N: int,
              bias: Optional[torch.Tensor] = None, epilogue: int = _lib.EPI_BIAS, groups: int = 1,
              res: Optional[torch.Tensor] = None, gamma_beta: Optional[torch.Tensor] = None, gb_shared: bool = False,
              gb_half: int = 0, pos_table: Optional[torch.Tensor] = None, lengths: Optional[torch.Tensor] = None,
              shift_by_group: bool = False, a_grouped: bool = True, norm_out: Optional[torch.Tensor] = None,
              norm_D: int = 0, norm_gamma: Optional[torch.Tensor] = None, norm_gb: Optional[torch.Tensor] = None,
              norm_gb_shared: bool = False, norm_gb_half: int = 0, tile: int = 0, taps_inner: Optional[bool] = None,
              norm_ssq: Optional[torch.Tensor] = None, row_ssq: Optional[torch.Tensor] = None, row_D: int = 0,
              row_bias: Optional[torch.Tensor] = None, row_bias_shared: bool = False, a_kblocked: bool = False,
              w_kblocked: bool = False, out_kblocked: bool = False, band: int = 0, x3: bool = False,
              shared_rows: Optional[bool] = None, pre_out: Optional[torch.Tensor] = None):
    """out = epilogue(sum_terms shift(A) @ W^T).

    terms: (A [G?,M,lda], W [G?,Np,K], shift).  With groups > 1 the leading dim of A (unless
    a_grouped=False), W, bias, out, res and gamma_beta ([Bc, G, 2*half]) is the group.
    x3: DN_BF16X3 -- A and W are split rows (packing.split_rows: bf16 tensors of twice the element count per row); a bf16 `out` /
    `norm_out` is then written as split rows too, an fp32 one plainly.
    """
    lib = _lib.load()
    p = _lib.GemmParams()
    A0, W0, _ = terms[0]
    M = A0.shape[-2]
    p.n_terms = len(terms)
    p.dtype = _lib.DN_BF16X3 if x3 else _code(A0)
    sw = 2 if x3 else 1  # a split row stores 2 bf16 per element
    assert not x3 or (A0.dtype == torch.bfloat16 and not (a_kblocked or w_kblocked or out_kblocked))
    code_o = lambda t: _lib.DN_BF16X3 if (x3 and t.dtype == torch.bfloat16) else _code(t)
    wid = lambda t: t.shape[-1] // 2 if (x3 and t.dtype == torch.bfloat16) else t.shape[-1]
    # K-blocked operands (packing.kblock): [K/32, rows, 32]
    K = W0.shape[-3] * 32 if w_kblocked else W0.shape[-1] // sw
    p.M, p.N, p.K, p.T = M, N, K, T
    p.groups, p.epilogue = groups, epilogue
    for i, (A, W, shift) in enumerate(terms):
        assert A.is_contiguous() and W.is_contiguous() and A.dtype == W.dtype
        t = p.terms[i]
        t.A, t.W, t.lda, t.shift = A.data_ptr(), W.data_ptr(), (K if a_kblocked else A.shape[-1] // sw), shift
        t.layout = (_lib.LAYOUT_A_KBLOCKED if a_kblocked else 0) | (_lib.LAYOUT_W_KBLOCKED if w_kblocked else 0)
        t.a_gstride = A.shape[-2] * A.shape[-1] // sw if (groups > 1 and a_grouped and A.dim() == 3) else 0
        t.w_gstride = W.shape[-2] * W.shape[-1] // sw if (groups > 1 and W.dim() == 3) else 0
        t.shift_by_group = int(shift_by_group)
    if bias is not None:
        assert bias.dtype == torch.float32
        p.bias = bias.data_ptr()
        p.bias_gstride = bias.shape[-1] if groups > 1 else 0
    p.out, p.ldo, p.out_dtype = out.data_ptr(), (N if out_kblocked else wid(out)), code_o(out)
    p.out_layout = _lib.LAYOUT_OUT_KBLOCKED if out_kblocked else 0
    p.out_gstride = (M * N if out_kblocked else out.shape[-2] * wid(out)) if groups > 1 else 0
    if res is not None:
        p.res, p.ldr, p.res_dtype = res.data_ptr(), res.shape[-1], _code(res)
        p.res_gstride = res.shape[-2] * res.shape[-1] if groups > 1 else 0
    if gamma_beta is not None:
        assert gamma_beta.dtype == torch.float32
        p.gamma_beta = gamma_beta.data_ptr()
        p.gb_ld = 0 if gb_shared else gamma_beta.stride(0)
        p.gb_half = gb_half
        p.gb_gstride = 2 * gb_half if groups > 1 else 0
    if pos_table is not None:
        p.pos_table, p.pos_ld = pos_table.data_ptr(), pos_table.shape[-1]
        p.lengths = lengths.data_ptr()
    if norm_out is not None:  # fused RMSNorm of the produced rows (RESADD / POSEMB, N <= 512)
        p.norm_out, p.norm_ld, p.norm_dtype, p.norm_D = norm_out.data_ptr(), wid(norm_out), code_o(norm_out), norm_D
        p.norm_gamma = _lib.ptr(norm_gamma)
        p.norm_gb = _lib.ptr(norm_gb)
        p.norm_gb_ld = 0 if (norm_gb is None or norm_gb_shared) else norm_gb.stride(0)
        p.norm_gb_half = norm_gb_half
        if norm_ssq is not None:  # split norm, producer side: norm_out = row * gamma, norm_ssq = partial sums of squares
            assert norm_ssq.dtype == torch.float32 and norm_ssq.shape[-1] * 64 >= N
            p.norm_split, p.norm_ssq, p.norm_ssq_ld = 1, norm_ssq.data_ptr(), norm_ssq.shape[-1]
    if row_ssq is not None:  # split norm, consumer side
        assert row_ssq.dtype == torch.float32 and row_D > 0
        p.row_ssq, p.row_ssq_ld, p.row_ssq_parts, p.row_D = row_ssq.data_ptr(), row_ssq.shape[-1], row_ssq.shape[-1], float(row_D)
        if row_bias is not None:
            assert row_bias.dtype == torch.float32
            p.row_bias = row_bias.data_ptr()
            p.row_bias_ld = 0 if row_bias_shared else row_bias.stride(0)
    if pre_out is not None:  # GEGLU: the pre-activation (packed columns) kept for the backward pass
        assert pre_out.is_contiguous() and pre_out.dtype == out.dtype
        p.pre_out, p.pre_ld = pre_out.data_ptr(), pre_out.shape[-1]
    p.pad_ = int(os.environ.get("DN_DEBUG_FLAGS", "0")) | (tile << 16) | (0 if taps_inner is None else (1 << 22) if taps_inner else (1 << 23)) | ((band & 0x7f) << 24) | ((1 << 21) if shared_rows is False else 0)  # ablation switches (tools/gemm_bench.py) | forced tile / K order / no shared staging of the taps' rows (tests; None = the library's default)
    _lib.check(lib.dn_conv_gemm(C.byref(p), _stream()), "dn_conv_gemm")
    return out


def attention(q, k, v, out, B, T, heads, dim_head, lengths: Optional[torch.Tensor], ldq=None, ldk=None, ldv=None):
    lib = _lib.load()
    a = _lib.AttnParams()
    a.q, a.k, a.v, a.out = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    a.ldq, a.ldk, a.ldv, a.ldo = ldq or q.shape[-1], ldk or k.shape[-1], ldv or v.shape[-1], out.shape[-1]
    a.B, a.T, a.heads, a.dim_head = B, T, heads, dim_head
    a.dtype = _code(q)
    a.lengths = _lib.ptr(lengths)
    a.scale = dim_head ** -0.5
    _lib.check(lib.dn_attention(C.byref(a), _stream()), "dn_attention")
    return out


def rmsnorm(x, out, T, D=None, gamma=None, gamma_beta=None, gb_shared=False, gb_half=0):
    lib = _lib.load()
    M = x.numel() // x.shape[-1]
    D = D or x.shape[-1]
    gb_ld = 0 if (gamma_beta is None or gb_shared) else gamma_beta.stride(0)
    _lib.check(lib.dn_rmsnorm(x.data_ptr(), x.shape[-1], out.data_ptr(), out.shape[-1], _code(out), M, D, T,
                              _lib.ptr(gamma), _lib.ptr(gamma_beta), gb_ld, gb_half, _stream()), "dn_rmsnorm")
    return out


def time_cond(times_i32, w_freq, W, bias, out, out_act=None):
    lib = _lib.load()
    B, Cn = times_i32.numel(), W.shape[0]
    _lib.check(lib.dn_time_cond(times_i32.data_ptr(), B, w_freq.data_ptr(), w_freq.numel(), W.data_ptr(), bias.data_ptr(), Cn,
                                out.data_ptr(), _lib.ptr(out_act), _code(out_act) if out_act is not None else DN_F32,
                                out.shape[-1], _stream()), "dn_time_cond")
    return out


def ddim_step(x, eps, coef, t_i32, T, out=None):
    """x, eps fp32 [B,T,C] dense -> x_prev (reference latent_module.py:1419-1442)."""
    lib = _lib.load()
    out = torch.empty_like(x) if out is None else out
    C_ = x.shape[-1]
    M = x.numel() // C_
    _lib.check(lib.dn_ddim_step(x.data_ptr(), eps.data_ptr(), out.data_ptr(), None, DN_F32, C_, M, C_, C_, T, coef.data_ptr(),
                                t_i32.data_ptr(), _stream()), "dn_ddim_step")
    return out


def q_sample(x, noise, coef_a, coef_b, t_i32, T, out=None):
    """out = a[t]*x + b[t]*noise on fp32 [B,T,C] dense tensors."""
    lib = _lib.load()
    out = torch.empty_like(x) if out is None else out
    C_ = x.shape[-1]
    M = x.numel() // C_
    _lib.check(lib.dn_q_sample(x.data_ptr(), noise.data_ptr(), out.data_ptr(), None, DN_F32, C_, M, C_, C_, T, coef_a.data_ptr(),
                               coef_b.data_ptr(), t_i32.data_ptr(), _stream()), "dn_q_sample")
    return out


def argmax_units(logits, offset=4):
    lib = _lib.load()
    V = logits.shape[-1]
    M = logits.numel() // V
    units = torch.empty(logits.shape[:-1], dtype=torch.int32, device=logits.device)
    _lib.check(lib.dn_argmax_units(logits.data_ptr(), V, M, V, offset, units.data_ptr(), _stream()), "dn_argmax_units")
    return units


def randn(shape, seed: int, offset: int = 0, device="cuda:0"):
    lib = _lib.load()
    out = torch.empty(shape, dtype=torch.float32, device=device)
    _lib.check(lib.dn_randn(out.data_ptr(), out.numel(), seed, offset, _stream()), "dn_randn")
    return out


def convert_rows(src, dst, C_):
    lib = _lib.load()
    M = src.numel() // src.shape[-1]
    _lib.check(lib.dn_convert_rows(src.data_ptr(), _code(src), src.shape[-1], dst.data_ptr(), _code(dst), dst.shape[-1], M, C_,
                                   _stream()), "dn_convert_rows")
    return dst


def conv_weight_grad_tn(x: torch.Tensor, dy: torch.Tensor, T: int, cin: int, cout: int, shifts: Sequence[int],
                        slices: int = 1) -> torch.Tensor:
    """conv_weight_grad without transposed operand copies (dn_conv_weight_grad_tn, bf16): the contraction over frames reads the
    row-major x / dy through transposing LDS reads.  x [B*T, ldx], dy [B*T, lddy] with zero pad columns.  -> fp32 [taps, cout, cin]."""
    import ctypes as C_

    lib = _lib.load()
    assert x.dtype == dy.dtype == torch.bfloat16 and x.dim() == 2 and dy.dim() == 2 and x.is_contiguous() and dy.is_contiguous()
    M = x.shape[0]
    B = M // T
    n_taps = len(shifts)
    rows_w, Np, Kp = (cin + 127) // 128 * 128, (cout + 127) // 128 * 128, (cin + 63) // 64 * 64
    xs = (C_.c_void_p * n_taps)(*([x.data_ptr()] * n_taps))
    ld = (C_.c_int32 * n_taps)(*([x.shape[1]] * n_taps))
    sh = (C_.c_int32 * n_taps)(*[int(s_) for s_ in shifts])
    grad = torch.zeros((n_taps, Np, Kp), device=x.device, dtype=torch.float32)
    if slices == 1:
        _lib.check(lib.dn_conv_weight_grad_tn(dy.data_ptr(), dy.shape[1], cout, xs, ld, sh, n_taps, cin, B, T, 1, None, grad.data_ptr(), _stream()),
                   "dn_conv_weight_grad_tn")
    else:
        part = torch.empty((slices, cout, n_taps * rows_w), device=x.device, dtype=torch.float32)
        _lib.check(lib.dn_conv_weight_grad_tn(dy.data_ptr(), dy.shape[1], cout, xs, ld, sh, n_taps, cin, B, T, slices, part.data_ptr(), None, _stream()),
                   "dn_conv_weight_grad_tn")
        _lib.check(lib.dn_wgrad_reduce(part.data_ptr(), slices, cout, n_taps * rows_w, rows_w, n_taps, grad.data_ptr(), Np, Kp, _stream()), "dn_wgrad_reduce")
    return grad[:, :cout, :cin]


def conv_weight_grad(x: torch.Tensor, dy: torch.Tensor, T: int, cin: int, cout: int, shifts: Sequence[int],
                     k_slices: int = 0) -> torch.Tensor:
    """Weight gradient of a causal conv / Linear, y[t] = sum_j W_j x[t - shifts[j]]: dW_j = sum_frames
    dy[t] (x) x[t - shifts[j]] -> fp32 [len(shifts), cout, cin].  x [B*T, >= cin], dy [B*T, >= cout], both bf16 or both fp32.

    The contraction over frames runs on the same dn_conv_gemm as the forward: both operands are transposed to channels-major
    (dn_transpose_pad; x once per tap with `shift` zero frames in front of every sequence), the frame index is split into
    k_slices groups that fill the chip, and the fp32 partial sums are added at the end."""
    lib = _lib.load()
    assert x.dtype == dy.dtype and x.dtype in (torch.bfloat16, torch.float32) and x.dim() == 2 and dy.dim() == 2
    f32 = x.dtype == torch.float32
    M = x.shape[0]
    assert M % T == 0 and dy.shape[0] == M and all(s >= 0 for s in shifts)
    B = M // T
    q = 64 * max(k_slices, 1)  # an explicit slice count must divide the padded frame index into whole K-tiles
    Tp = (T + max(shifts) + q - 1) // q * q
    cols = B * Tp
    if k_slices <= 0:  # enough 256-row tiles for the whole chip, K-slices of at least 8 K-tiles
        tiles = ((cout + 255) // 256) * ((cin + 255) // 256)
        k_slices = 1
        while k_slices * 2 * tiles <= 512 and cols % (k_slices * 2 * 64) == 0 and cols // (k_slices * 2) >= 512:
            k_slices *= 2
    chunk = cols // k_slices
    assert chunk * k_slices == cols and chunk % 64 == 0
    rows_a, rows_w = cout, (cin + 127) // 128 * 128  # A rows are clamped by the kernel, packed-weight rows go in 128s
    n_taps = len(shifts)
    tp = lib.dn_transpose_pad_f32 if f32 else lib.dn_transpose_pad

    def transposed(src, C_, front, dst, rows, row0):
        _lib.check(tp(src.data_ptr(), src.shape[1], B, T, C_, front, Tp, dst.data_ptr(), rows, dst.shape[1], row0, chunk,
                      _stream()), "dn_transpose_pad")

    dyT = torch.empty((k_slices, rows_a, chunk), device=x.device, dtype=x.dtype)
    transposed(dy, cout, 0, dyT, rows_a, 0)
    xT = torch.empty((k_slices, n_taps * rows_w, chunk), device=x.device, dtype=x.dtype)  # the taps' operands, stacked
    for j, s in enumerate(shifts):
        transposed(x, cin, s, xT, rows_w, j * rows_w)
    N = n_taps * rows_w
    part = torch.empty((k_slices, cout, N), device=x.device, dtype=torch.float32)
    conv_gemm([(dyT, xT, 0)], part, cout, N, groups=k_slices)  # one contraction for all taps: dY^T is read once
    Np, Kp = (cout + 127) // 128 * 128, (cin + 63) // 64 * 64
    grad = torch.zeros((n_taps, Np, Kp), device=x.device, dtype=torch.float32)  # the packed layout of the weight itself
    _lib.check(lib.dn_wgrad_reduce(part.data_ptr(), k_slices, cout, N, rows_w, n_taps, grad.data_ptr(), Np, Kp, _stream()),
               "dn_wgrad_reduce")
    return grad[:, :cout, :cin].contiguous()


# ------------------------------------------------------------------------------------------ backward ops
def _scratch(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=device)


def attention_fwd_lse(q, k, v, out, B, T, heads, dim_head, lengths, ldq=None, ldk=None, ldv=None, dropout_p=0.0, seed=0):
    """dn_attention that also keeps the per-query log-sum-exp [B, heads, T] for `attention_backward`.  dropout_p > 0: train-mode
    dropout on the probabilities with the counter-hash mask of `seed` (64 bits)."""
    lib = _lib.load()
    lse = torch.empty(B, heads, T, dtype=torch.float32, device=q.device)
    a = _lib.AttnParams()
    a.q, a.k, a.v, a.out = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    a.ldq, a.ldk, a.ldv, a.ldo = ldq or q.shape[-1], ldk or k.shape[-1], ldv or v.shape[-1], out.shape[-1]
    a.B, a.T, a.heads, a.dim_head = B, T, heads, dim_head
    a.dtype = _code(q)
    a.lengths = _lib.ptr(lengths)
    a.scale = dim_head ** -0.5
    a.lse = lse.data_ptr()
    a.dropout_p, a.seed_lo, a.seed_hi = float(dropout_p), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    _lib.check(lib.dn_attention(C.byref(a), _stream()), "dn_attention")
    return out, lse


def attention_backward(q, k, v, out, dout, lse, B, T, heads, dim_head, lengths, ld_qkv=None, dropout_p=0.0, seed=0):
    """-> (dq, dk, dv) as three column blocks of one [B*T, 3*heads*dim_head] tensor (reference :299-343 differentiated)."""
    lib = _lib.load()
    hd = heads * dim_head
    dqkv = torch.empty(B * T, 3 * hd, dtype=q.dtype, device=q.device)
    delta = torch.empty(B, heads, T, dtype=torch.float32, device=q.device)
    a = _lib.AttnBwdParams()
    a.q, a.k, a.v, a.out, a.dout = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), dout.data_ptr()
    es = dqkv.element_size()
    a.dq, a.dk, a.dv = dqkv.data_ptr(), dqkv.data_ptr() + hd * es, dqkv.data_ptr() + 2 * hd * es
    ld = ld_qkv or q.shape[-1]
    a.ldq = a.ldk = a.ldv = ld
    a.ldo, a.lddo = out.shape[-1], dout.shape[-1]
    a.lddq = a.lddk = a.lddv = 3 * hd
    a.B, a.T, a.heads, a.dim_head = B, T, heads, dim_head
    a.dtype = _code(q)
    a.lengths = _lib.ptr(lengths)
    a.scale = dim_head ** -0.5
    a.lse, a.delta = lse.data_ptr(), delta.data_ptr()
    a.dropout_p, a.seed_lo, a.seed_hi = float(dropout_p), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    _lib.check(lib.dn_attention_backward(C.byref(a), _stream()), "dn_attention_backward")
    return dqkv[:, :hd], dqkv[:, hd:2 * hd], dqkv[:, 2 * hd:]


def gate_forward(h, res, T, gamma_beta=None, gb_half=0):
    lib = _lib.load()
    out = torch.empty_like(h)
    M, ld = h.shape
    _lib.check(lib.dn_gate_forward(h.data_ptr(), res.data_ptr(), out.data_ptr(), _code(h), M, ld, T, _lib.ptr(gamma_beta),
                                   gamma_beta.stride(0) if gamma_beta is not None else 0, gb_half, _stream()), "dn_gate_forward")
    return out


def gate_backward(dout, h, T, gamma_beta=None, gb_half=0, want_rows=False):
    lib = _lib.load()
    dh = torch.empty_like(h)
    M, ld = h.shape
    rows = torch.zeros(M, 2 * gb_half, dtype=torch.float32, device=h.device) if want_rows else None
    _lib.check(lib.dn_gate_backward(dout.data_ptr(), h.data_ptr(), dh.data_ptr(), _code(h), M, ld, T, _lib.ptr(gamma_beta),
                                    gamma_beta.stride(0) if gamma_beta is not None else 0, gb_half, _lib.ptr(rows),
                                    2 * gb_half, _stream()), "dn_gate_backward")
    return (dh, rows) if want_rows else dh


def geglu_forward(pre, ip):
    lib = _lib.load()
    M = pre.shape[0]
    out = torch.empty(M, ip, dtype=pre.dtype, device=pre.device)
    _lib.check(lib.dn_geglu_forward(pre.data_ptr(), out.data_ptr(), _code(pre), M, ip, _stream()), "dn_geglu_forward")
    return out


def geglu_backward(dout, pre, ip):
    lib = _lib.load()
    dpre = torch.empty_like(pre)
    _lib.check(lib.dn_geglu_backward(dout.data_ptr(), pre.data_ptr(), dpre.data_ptr(), _code(pre), pre.shape[0], ip, _stream()),
               "dn_geglu_backward")
    return dpre


def rmsnorm_backward(x, dy, B, T, D, gamma=None, gamma_beta=None, gb_half=0, dres=None, act_dtype=None, dgamma=None, dgamma_beta=None):
    """-> (dx fp32 [B*T, ldx], dx_act or None); dgamma / dgamma_beta are accumulated in place."""
    lib = _lib.load()
    dx = torch.empty_like(x)
    dx_act = torch.empty(x.shape[0], x.shape[1], dtype=act_dtype, device=x.device) if act_dtype is not None else None
    scratch = _scratch(int(lib.dn_rmsnorm_backward_scratch_bytes(B, T, D)), x.device)
    _lib.check(lib.dn_rmsnorm_backward(x.data_ptr(), x.shape[1], dy.data_ptr(), dy.shape[1], _code(dy), B, T, D, _lib.ptr(gamma),
                                       _lib.ptr(gamma_beta), gamma_beta.stride(0) if gamma_beta is not None else 0, gb_half,
                                       _lib.ptr(dres), dx.data_ptr(), _lib.ptr(dx_act), _code(dx_act) if dx_act is not None else DN_F32,
                                       x.shape[1], _lib.ptr(dgamma), _lib.ptr(dgamma_beta),
                                       dgamma_beta.stride(0) if dgamma_beta is not None else 0, scratch.data_ptr(), _stream()),
               "dn_rmsnorm_backward")
    return dx, dx_act


def colsum(src, groups, rows_per_group, C_, out=None, scale=1.0, accumulate=False):
    lib = _lib.load()
    out = torch.zeros(groups, C_, dtype=torch.float32, device=src.device) if out is None else out
    scratch = _scratch(int(lib.dn_colsum_scratch_bytes(groups, rows_per_group, C_)), src.device)
    _lib.check(lib.dn_colsum(src.data_ptr(), src.shape[-1], _code(src), groups, rows_per_group, C_, out.data_ptr(), out.stride(0),
                             float(scale), int(accumulate), scratch.data_ptr(), _stream()), "dn_colsum")
    return out


def posterior_backward(params, noise, dz, Z, T, lengths, kl_weight, act_dtype, ldo):
    lib = _lib.load()
    M = params.shape[0]
    out = torch.empty(M, ldo, dtype=act_dtype, device=params.device)
    _lib.check(lib.dn_posterior_backward(params.data_ptr(), params.shape[1], noise.data_ptr(), noise.shape[1], dz.data_ptr(), dz.shape[1],
                                         out.data_ptr(), _code(out), ldo, M, Z, T, _lib.ptr(lengths), float(kl_weight), _stream()),
               "dn_posterior_backward")
    return out


def lsce_loss_grad(logits, target_i32, epsilon, grad_scale, act_dtype=None, ldd=0):
    """-> (rows [M,4] = nll, smooth, correct, valid; dlogits [M, ldd] or None)."""
    lib = _lib.load()
    M, V = logits.shape
    rows = torch.empty(M, 4, dtype=torch.float32, device=logits.device)
    dl = torch.empty(M, ldd, dtype=act_dtype, device=logits.device) if act_dtype is not None else None
    _lib.check(lib.dn_lsce_loss_grad(logits.data_ptr(), logits.stride(0), target_i32.data_ptr(), M, V, float(epsilon), float(grad_scale),
                                     rows.data_ptr(), _lib.ptr(dl), _code(dl) if dl is not None else DN_F32, ldd, _stream()),
               "dn_lsce_loss_grad")
    return rows, dl


def masked_mse_grad(pred, target, T, lengths, grad_scale, C_, dpred=None, accumulate=False, act_dtype=None, ld_act=0):
    lib = _lib.load()
    M = pred.shape[0]
    sq = torch.empty(M, dtype=torch.float32, device=pred.device)
    dact = torch.empty(M, ld_act, dtype=act_dtype, device=pred.device) if act_dtype is not None else None
    _lib.check(lib.dn_masked_mse_grad(pred.data_ptr(), pred.stride(0), target.data_ptr(), target.stride(0), M, C_, T, _lib.ptr(lengths),
                                      float(grad_scale), sq.data_ptr(), _lib.ptr(dpred), dpred.stride(0) if dpred is not None else 0,
                                      int(accumulate), _lib.ptr(dact), _code(dact) if dact is not None else DN_F32, ld_act, _stream()),
               "dn_masked_mse_grad")
    return sq, dact


def sum_groups(src):
    """[count, ...] -> sum over the leading dim."""
    lib = _lib.load()
    dst = torch.empty_like(src[0])
    _lib.check(lib.dn_sum_groups(src.data_ptr(), src[0].numel(), src.shape[0], dst.data_ptr(), _code(src), dst.numel(), _stream()),
               "dn_sum_groups")
    return dst


def transpose_weights(w, Rp, Cp):
    """[count, R, Cc] -> [count, Cp, Rp] padded transposes."""
    lib = _lib.load()
    n, R, Cc = w.shape
    out = torch.empty(n, Cp, Rp, dtype=w.dtype, device=w.device)
    _lib.check(lib.dn_transpose_weights(w.data_ptr(), _code(w), n, R * Cc, R, Cc, out.data_ptr(), Rp * Cp, Rp, Cp, _stream()),
               "dn_transpose_weights")
    return out
