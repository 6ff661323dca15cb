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
    raise TypeError(f"unsupported tensor dtype {t.dtype}")


def _stream():
    return _lib.current_stream()


def conv_gemm(terms: Sequence[Tuple[torch.Tensor, torch.Tensor, int]], out: torch.Tensor, T: int, N: int,
              bias: Optional[torch.Tensor] = None, epilogue: int = _lib.EPI_BIAS, groups: int = 1,
              res: Optional[torch.Tensor] = None, gamma_beta: Optional[torch.Tensor] = None, gb_shared: bool = False,
              gb_half: int = 0, pos_table: Optional[torch.Tensor] = None, lengths: Optional[torch.Tensor] = None,
              shift_by_group: bool = False, a_grouped: bool = True, norm_out: Optional[torch.Tensor] = None,
              norm_D: int = 0, norm_gamma: Optional[torch.Tensor] = None, norm_gb: Optional[torch.Tensor] = None,
              norm_gb_shared: bool = False, norm_gb_half: int = 0, tile: int = 0, taps_inner: bool = False,
              norm_ssq: Optional[torch.Tensor] = None, row_ssq: Optional[torch.Tensor] = None, row_D: int = 0,
              row_bias: Optional[torch.Tensor] = None, row_bias_shared: bool = False, a_kblocked: bool = False,
              w_kblocked: bool = False, out_kblocked: bool = False):
    """out = epilogue(sum_terms shift(A) @ W^T).

    terms: (A [G?,M,lda], W [G?,Np,K], shift).  With groups > 1 the leading dim of A (unless
    a_grouped=False), W, bias, out, res and gamma_beta ([Bc, G, 2*half]) is the group.
    """
    lib = _lib.load()
    p = _lib.GemmParams()
    A0, W0, _ = terms[0]
    M = A0.shape[-2]
    p.n_terms = len(terms)
    p.dtype = _code(A0)
    # K-blocked operands (packing.kblock): [K/32, rows, 32]
    K = W0.shape[-3] * 32 if w_kblocked else W0.shape[-1]
    p.M, p.N, p.K, p.T = M, N, K, T
    p.groups, p.epilogue = groups, epilogue
    for i, (A, W, shift) in enumerate(terms):
        assert A.is_contiguous() and W.is_contiguous() and A.dtype == W.dtype
        t = p.terms[i]
        t.A, t.W, t.lda, t.shift = A.data_ptr(), W.data_ptr(), (K if a_kblocked else A.shape[-1]), shift
        t.layout = (_lib.LAYOUT_A_KBLOCKED if a_kblocked else 0) | (_lib.LAYOUT_W_KBLOCKED if w_kblocked else 0)
        t.a_gstride = A.shape[-2] * A.shape[-1] if (groups > 1 and a_grouped and A.dim() == 3) else 0
        t.w_gstride = W.shape[-2] * W.shape[-1] if (groups > 1 and W.dim() == 3) else 0
        t.shift_by_group = int(shift_by_group)
    if bias is not None:
        assert bias.dtype == torch.float32
        p.bias = bias.data_ptr()
        p.bias_gstride = bias.shape[-1] if groups > 1 else 0
    p.out, p.ldo, p.out_dtype = out.data_ptr(), (N if out_kblocked else out.shape[-1]), _code(out)
    p.out_layout = _lib.LAYOUT_OUT_KBLOCKED if out_kblocked else 0
    p.out_gstride = (M * N if out_kblocked else out.shape[-2] * out.shape[-1]) if groups > 1 else 0
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
        p.norm_out, p.norm_ld, p.norm_dtype, p.norm_D = norm_out.data_ptr(), norm_out.shape[-1], _code(norm_out), norm_D
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
    p.pad_ = int(os.environ.get("DN_DEBUG_FLAGS", "0")) | (tile << 16) | (int(taps_inner) << 22)  # ablation switches (tools/gemm_bench.py) | forced tile / K order (tests)
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


def conv_weight_grad(x: torch.Tensor, dy: torch.Tensor, T: int, cin: int, cout: int, shifts: Sequence[int],
                     k_slices: int = 0) -> torch.Tensor:
    """Weight gradient of a causal conv / Linear, y[t] = sum_j W_j x[t - shifts[j]]: dW_j = sum_frames
    dy[t] (x) x[t - shifts[j]] -> fp32 [len(shifts), cout, cin].  x [B*T, >= cin], dy [B*T, >= cout] bf16.

    The contraction over frames runs on the same dn_conv_gemm as the forward: both operands are transposed to channels-major
    (dn_transpose_pad; x once per tap with `shift` zero frames in front of every sequence), the frame index is split into
    k_slices groups that fill the chip, and the fp32 partial sums are added at the end."""
    lib = _lib.load()
    assert x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16 and x.dim() == 2 and dy.dim() == 2
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

    def transposed(src, C_, front, dst, rows, row0):
        _lib.check(lib.dn_transpose_pad(src.data_ptr(), src.shape[1], B, T, C_, front, Tp, dst.data_ptr(), rows, dst.shape[1], row0, chunk,
                                        _stream()), "dn_transpose_pad")

    dyT = torch.empty((k_slices, rows_a, chunk), device=x.device, dtype=torch.bfloat16)
    transposed(dy, cout, 0, dyT, rows_a, 0)
    xT = torch.empty((k_slices, n_taps * rows_w, chunk), device=x.device, dtype=torch.bfloat16)  # the taps' operands, stacked
    for j, s in enumerate(shifts):
        transposed(x, cin, s, xT, rows_w, j * rows_w)
    N = n_taps * rows_w
    part = torch.empty((k_slices, cout, N), device=x.device, dtype=torch.float32)
    conv_gemm([(dyT, xT, 0)], part, cout, N, groups=k_slices)  # one contraction for all taps: dY^T is read once
    return part.sum(dim=0).view(cout, n_taps, rows_w)[:, :, :cin].permute(1, 0, 2).contiguous()
