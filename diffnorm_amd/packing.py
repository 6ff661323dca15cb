"""State-dict -> packed device tensors for libdiffnorm_hip.so.

Input: tensors in the reference's state-dict layout (conv weights [Cout,Cin,k], linear
weights [out,in]).  Output: the ordered tensor tables `dn_eps_create` / `dn_vae_create` expect
(diffnorm_amd/csrc/engine.h documents each entry).  Packing rules:

* every weight becomes [rows padded to 128][K padded to 64] with K contiguous, in the arithmetic
  dtype (bf16, fp32, or DN_BF16X3 split rows: hi / lo bf16 halves per 32 elements, `split_rows`); pads are zeros, so padded channels stay exactly zero through the network;
* a k-tap causal conv becomes k matrices, tap j multiplying the frame t-(k-1-j)*dilation;
* Linear(D, 2*inner) of the GEGLU is interleaved per 16-row MFMA tile = [8 value rows ; 8 gate rows of the same output
  columns], so every tile is self-contained and one wave holds value and gate of a column (epilogue DN_EPI_GEGLU);
* the 2*S*L FiLM and 2*depth adaptive-RMSNorm projections are stacked into one [n_cond, C] matrix,
  each as [gamma(Dp) ; beta(Dp)];
* biases, norm gammas, the Fourier frequencies and the sinusoidal table stay fp32.
"""
import math
from typing import Dict, List

import torch

from . import _lib

SD = Dict[str, torch.Tensor]


def padk(c: int) -> int:
    return (c + 63) // 64 * 64


def padn(c: int) -> int:
    return (c + 127) // 128 * 128


def _act_dtype(dtype: int):
    return torch.bfloat16 if dtype == _lib.DN_BF16 else torch.float16 if dtype == _lib.DN_F16 else torch.float32


def _is16(dtype: int) -> bool:
    """the two 2-byte arithmetic modes share every packed layout (K-blocked copies included)"""
    return dtype in (_lib.DN_BF16, _lib.DN_F16)


def split_rows(t: torch.Tensor, weight: bool = False) -> torch.Tensor:
    """fp32 [..., K] (K a multiple of 32) -> DN_BF16X3 "split rows" (include/diffnorm_hip.h): a bf16 tensor [..., 2 K] of the same
    byte size as the fp32 one, every group of 32 elements stored as two 64-byte halves: hi = bf16(x) and lo = bf16(x - hi) (the
    difference is exact in fp32), so x = hi + lo to 16 mantissa bits.  Activations store [hi | lo], weights [lo | hi]: a
    contraction that walks a row in 64-byte K-tiles then meets (w_lo, a_hi) and (w_hi, a_lo) and has to keep only the
    activations' hi fragments across the pair for its three products w_lo a_hi + w_hi a_lo + w_hi a_hi."""
    t = t.float()
    assert t.shape[-1] % 32 == 0, t.shape
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    halves = [lo, hi] if weight else [hi, lo]
    g = torch.stack([h.reshape(*t.shape[:-1], -1, 32) for h in halves], dim=-2)  # [..., K/32, 2, 32]
    return g.reshape(*t.shape[:-1], 2 * t.shape[-1]).contiguous()


def unsplit_rows(t: torch.Tensor) -> torch.Tensor:
    """Inverse of split_rows (either order; tests): bf16 [..., 2 K] -> fp32 [..., K] = hi + lo."""
    g = t.reshape(*t.shape[:-1], -1, 2, 32).float()
    return (g[..., 0, :] + g[..., 1, :]).reshape(*t.shape[:-1], t.shape[-1] // 2)


def _arith(t: torch.Tensor, dtype: int, weight: bool = True) -> torch.Tensor:
    """fp32 tensor (last dim = K, padded) -> the arithmetic dtype's storage (weight: a packed weight matrix, else activation rows)."""
    if dtype == _lib.DN_BF16X3:
        return split_rows(t, weight=weight)
    if dtype == _lib.DN_F16:  # the format ends at 65504: saturate, as the kernels do (FP16_OVFL), instead of inf
        return t.float().clamp(-65504.0, 65504.0).to(torch.float16)
    return t.to(_act_dtype(dtype))


def _mat(w: torch.Tensor, dtype: int, rows: int = None, cols: int = None) -> torch.Tensor:
    """[N,K] -> zero-padded [rows or padn(N), cols or padk(K)] in the arithmetic dtype."""
    n, k = w.shape
    out = torch.zeros(rows or padn(n), cols or padk(k), dtype=torch.float32)
    out[:n, :k] = w.float()
    return _arith(out, dtype)


def _vec(b: torch.Tensor, n: int) -> torch.Tensor:
    out = torch.zeros(n, dtype=torch.float32)
    out[: b.numel()] = b.float().flatten()
    return out


def _conv(w: torch.Tensor, dtype: int) -> torch.Tensor:
    """[Cout,Cin,k] -> [k, padn(Cout), padk(Cin)], tap j = w[:, :, j]."""
    return torch.stack([_mat(w[:, :, j], dtype) for j in range(w.shape[2])])


def sinusoidal_table(num: int, dim: int, ld: int) -> torch.Tensor:
    """fairseq SinusoidalPositionalEmbedding table with padding_idx 0
    (reference fairseq/modules/sinusoidal_positional_embedding.py:36-58): rows
    [sin(p f_j) | cos(p f_j)], f_j = exp(-j ln(1e4)/(dim/2-1)), row 0 = 0; built in fp32 like upstream."""
    half = dim // 2
    f = torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1)))
    ang = torch.arange(num, dtype=torch.float).unsqueeze(1) * f.unsqueeze(0)
    tab = torch.zeros(num, ld, dtype=torch.float32)
    tab[:, :half] = torch.sin(ang)
    tab[:, half:2 * half] = torch.cos(ang)
    tab[0] = 0
    return tab


def pack_wavenet(sd: SD, prefix: str, cin: int, cout: int, stacks: int, layers: int, dtype: int) -> List[torch.Tensor]:
    cp = padk(cout)
    g = lambda k: sd[prefix + k]
    conv_W, conv_b, res_W, res_b = [], [], [], []
    for s in range(stacks):
        for i in range(layers):
            p = f"stacks.{s}.blocks.{i}."
            conv_W.append(_conv(g(p + "conv.weight"), dtype))
            conv_b.append(_vec(g(p + "conv.bias"), cp))
            res_W.append(_mat(g(p + "res_conv.weight")[:, :, 0], dtype))
            res_b.append(_vec(g(p + "res_conv.bias"), cp))
    last = f"stacks.{stacks - 1}.blocks."
    skip_W = torch.stack([_mat(g(f"{last}{i}.skip_conv.weight")[:, :, 0], dtype) for i in range(layers)])
    skip_b = _vec(sum(g(f"{last}{i}.skip_conv.bias").float() for i in range(layers)), cp)
    return [
        _conv(g("init_conv.weight"), dtype), _vec(g("init_conv.bias"), cp),
        torch.stack(conv_W), torch.stack(conv_b), torch.stack(res_W), torch.stack(res_b),
        skip_W, skip_b,
        _mat(g("final_conv.weight")[:, :, 0], dtype), _vec(g("final_conv.bias"), cp),
        # conv_W and res_W once more, K-blocked, for the 256 x 256 tile (bf16 only; placeholders in f32 mode)
        kblock(torch.stack(conv_W)) if _is16(dtype) else torch.zeros(4),
        kblock(torch.stack(res_W)) if _is16(dtype) else torch.zeros(4),
    ]


def _geglu_rows(inner: int) -> torch.Tensor:
    """Source row in Linear(D,2*inner).weight for every packed row (or -1 for a zero row)."""
    ip = padk(inner)
    p = torch.arange(2 * ip)
    # every 16-row MFMA tile is self-contained: 8 value rows then the 8 gate rows of the same output columns, so a kernel
    # may cut the packed matrix at any multiple of 16 rows (the 256 x 352 tile gives a wave 176 of them)
    col = (p // 16) * 8 + (p % 8)
    is_gate = (p % 16) >= 8
    src = torch.where(is_gate, col + inner, col)
    return torch.where(col < inner, src, torch.full_like(src, -1))


def pack_transformer(sd: SD, prefix: str, dim: int, depth: int, heads: int, dim_head: int, dtype: int,
                     conditioned: bool) -> List[torch.Tensor]:
    inner = int(dim * 4 * 2 / 3)
    ip, Dp, hd = padk(inner), padk(dim), heads * dim_head
    rows = _geglu_rows(inner)
    keep = rows >= 0
    g = lambda k: sd[prefix + k]
    qkv, out, ffin, ffin_b, ffc, ffc_b, ffo, ffo_b, g1, g2 = ([] for _ in range(10))
    for l in range(depth):
        p = f"layers.{l}."
        qkv.append(_mat(torch.cat([g(p + "1.to_q.weight"), g(p + "1.to_kv.weight")], dim=0), dtype))
        out.append(_mat(g(p + "1.to_out.weight"), dtype))
        w, b = g(p + "5.0.weight").float(), g(p + "5.0.bias").float()
        wp = torch.zeros(2 * ip, Dp)
        bp = torch.zeros(2 * ip)
        wp[keep, :dim] = w[rows[keep]]
        bp[keep] = b[rows[keep]]
        ffin.append(_arith(wp, dtype))
        ffin_b.append(bp)
        ffc.append(_conv(g(p + "5.2.1.weight"), dtype))
        ffc_b.append(_vec(g(p + "5.2.1.bias"), ip))
        ffo.append(_mat(g(p + "5.3.weight"), dtype))
        ffo_b.append(_vec(g(p + "5.3.bias"), Dp))
        if not conditioned:
            g1.append(g(p + "0.gamma").float())
            g2.append(g(p + "4.gamma").float())
    dummy = torch.zeros(4, dtype=torch.float32)
    return [
        torch.stack(qkv), torch.stack(out), torch.stack(ffin), torch.stack(ffin_b), torch.stack(ffc),
        torch.stack(ffc_b), torch.stack(ffo), torch.stack(ffo_b),
        torch.stack(g1) if g1 else dummy, torch.stack(g2) if g2 else dummy.clone(),
        g("to_pred.0.gamma").float().clone(), _mat(g("to_pred.1.weight"), dtype),
        # the FFN conv weights once more, K-blocked, for the 256 x 352 tile (bf16 only; a placeholder in f32 mode)
        kblock(torch.stack(ffc)) if _is16(dtype) else dummy.clone(),
        # the GEGLU projection's packed weights K-blocked (its activations arrive K-blocked from the split norm's producer)
        kblock(torch.stack(ffin)) if _is16(dtype) else dummy.clone(),
        # and the q/kv projection's (layers >= 1 read the attention norm's output K-blocked from the previous layer's last contraction)
        kblock(torch.stack(qkv)) if _is16(dtype) else dummy.clone(),
    ]


def pack_eps(sd: SD, cfg, dtype: int, max_pos: int = 2048) -> List[torch.Tensor]:
    """cfg: object with dim, latent_dim, depth, heads, dim_head, wavenet_layers, wavenet_stacks, dim_cond_mult [, dim_prompt,
    num_latents_m, resampler_depth: the conditional variant, whose extra tensors (csrc/engine.h kEpsCondTensors) follow the table]."""
    D, Dp = cfg.dim, padk(cfg.dim)
    C = D * cfg.dim_cond_mult
    cond = getattr(cfg, "dim_prompt", 0) > 0
    C2 = 2 * C if cond else C  # [time cond | pooled-prompt cond] (reference latent_module.py:784, 852)
    cond_rows, cond_b = [], []

    def add_cond(wname, bname):
        w, b = sd[wname].float(), sd[bname].float()  # [2D, C2]: [gamma ; beta]
        blk = torch.zeros(2 * Dp, C2)
        bb = torch.zeros(2 * Dp)
        blk[:D], blk[Dp:Dp + D] = w[:D], w[D:]
        bb[:D], bb[Dp:Dp + D] = b[:D], b[D:]
        cond_rows.append(blk)
        cond_b.append(bb)

    for s in range(cfg.wavenet_stacks):
        for i in range(cfg.wavenet_layers):
            p = f"wavenet.stacks.{s}.blocks.{i}.to_time_cond."
            add_cond(p + "weight", p + "bias")
    for l in range(cfg.depth):
        for j in ((0, 2, 4) if cond else (0, 4)):  # attention norm, [cross-attention norm,] feed-forward norm
            p = f"transformer.layers.{l}.{j}.to_gamma_beta."
            add_cond(p + "weight", p + "bias")
    cond_W = _mat(torch.cat(cond_rows, dim=0), _lib.DN_F32)  # conditioning stays fp32 in every mode
    tensors = [
        sd["to_time_cond.0.weights"].float().clone(),
        sd["to_time_cond.1.weight"].float().contiguous().clone(),
        sd["to_time_cond.1.bias"].float().clone(),
        cond_W, torch.cat(cond_b),
        _mat(sd["init_conv.weight"][:, :, 0], dtype), _vec(sd["init_conv.bias"], Dp),
    ]
    wsd = {k[len("wavenet."):]: v for k, v in sd.items() if k.startswith("wavenet.")}
    tensors += pack_wavenet(wsd, "", D, D, cfg.wavenet_stacks, cfg.wavenet_layers, dtype)
    tsd = {k[len("transformer."):]: v for k, v in sd.items() if k.startswith("transformer.")}
    tensors += pack_transformer(tsd, "", D, cfg.depth, cfg.heads, cfg.dim_head, dtype, conditioned=True)
    tensors += [
        _mat(sd["final_proj.weight"], dtype), _vec(sd["final_proj.bias"], padk(cfg.latent_dim)),
        sinusoidal_table(max_pos + 1, D, Dp),
    ]
    if cond:
        tensors += pack_eps_cond(sd, cfg, dtype)
    return tensors


def pack_eps_cond(sd: SD, cfg, dtype: int) -> List[torch.Tensor]:
    """Extra tensors of the conditional variant in csrc/engine.h's kEpsCondTensors order (reference latent_module.py:416-471,
    752-773): prompt-condition MLP and null condition (fp32), null prompt tokens, the PerceiverResampler (latents with their
    sinusoidal positions 1..m folded in: a constant), and the transformer layers' cross-attention projections."""
    D, Dp, Dn = cfg.dim, padk(cfg.dim), padn(cfg.dim)
    C, P, m, R = D * cfg.dim_cond_mult, cfg.dim_prompt, cfg.num_latents_m, cfg.resampler_depth
    inner = int(D * 4 * 2 / 3)
    ip, hd = padk(inner), cfg.heads * cfg.dim_head
    rows = _geglu_rows(inner)
    keep = rows >= 0
    r = "perceiver_resampler."
    pad_rows = lambda t: torch.cat([t, torch.zeros(t.shape[0], Dp - t.shape[1])], dim=1)
    lat_pos = pad_rows(sd[r + "latents"].float() + sinusoidal_table(m + 1, D, D)[1:m + 1])
    rq, rkv, rout, rffin, rffin_b, rffout, rffout_b = ([] for _ in range(7))
    for l in range(R):
        a = f"{r}layers.{l}.0."
        f = f"{r}layers.{l}.1."
        rq.append(_mat(sd[a + "to_q.weight"], dtype))
        rkv.append(_mat(sd[a + "to_kv.weight"], dtype))
        rout.append(_mat(sd[a + "to_out.weight"], dtype))
        w, b = sd[f + "0.weight"].float(), sd[f + "0.bias"].float()
        wp, bp = torch.zeros(2 * ip, Dp), torch.zeros(2 * ip)
        wp[keep, :D], bp[keep] = w[rows[keep]], b[rows[keep]]
        rffin.append(_arith(wp, dtype))
        rffin_b.append(bp)
        rffout.append(_mat(sd[f + "2.weight"], dtype))
        rffout_b.append(_vec(sd[f + "2.bias"], Dp))
    t = "transformer.layers."
    return [
        _mat(sd["to_prompt_cond.1.weight"], _lib.DN_F32), sd["to_prompt_cond.1.bias"].float().clone(), sd["null_prompt_cond"].float().clone(),
        _arith(pad_rows(sd["null_prompt_tokens"].float()), dtype, weight=False),
        _mat(sd[r + "proj_context.weight"], dtype), _vec(sd[r + "proj_context.bias"], Dp), lat_pos,
        torch.stack(rq), torch.stack(rkv), torch.stack(rout), torch.stack(rffin), torch.stack(rffin_b), torch.stack(rffout),
        torch.stack(rffout_b), sd[r + "norm.gamma"].float().clone(),
        torch.stack([_mat(sd[f"{t}{l}.3.to_q.weight"], dtype) for l in range(cfg.depth)]),
        torch.stack([_mat(sd[f"{t}{l}.3.to_kv.weight"], dtype) for l in range(cfg.depth)]),
        torch.stack([_mat(sd[f"{t}{l}.3.to_out.weight"], dtype) for l in range(cfg.depth)]),
    ]


def vae_mults(latent_flag: int) -> List[int]:
    """chan_mults of SpeechVAEEncoderDecoder (reference latent_module.py:1044-1051)."""
    return {16: [4, 3, 2], 32: [4, 3], 128: [3]}[latent_flag]


def pack_vae(sd: SD, dim: int, mults: List[int], depth: int, heads: int, dim_head: int, stacks: int, layers: int,
             vocab: int, dtype: int) -> List[torch.Tensor]:
    tensors: List[torch.Tensor] = []
    cur = dim
    for n, m in enumerate(mults):
        tensors += pack_wavenet(sd, f"encoder_wave.{n}.", cur, cur // m, stacks, layers, dtype)
        cur //= m
    first = True
    for n, m in enumerate(reversed(mults)):
        tgt = cur * m
        cin = cur // 2 if first else cur
        first = False
        tensors += pack_wavenet(sd, f"decoder_wave.{n}.", cin, tgt, stacks, layers, dtype)
        cur = tgt
    tsd = {k[len("decoder_tf."):]: v for k, v in sd.items() if k.startswith("decoder_tf.")}
    tensors += pack_transformer(tsd, "", dim, depth, heads, dim_head, dtype, conditioned=False)
    tensors += [_mat(sd["decoder_lm.weight"], dtype), _vec(sd["decoder_lm.bias"], padn(vocab))]
    return tensors


def kblock(t: torch.Tensor) -> torch.Tensor:
    """[.., rows, K] -> the K-blocked layout [.., K/32, rows, 32] (include/diffnorm_hip.h, DN_LAYOUT_*): the 64 bytes a
    32-deep K-tile takes from a row sit next to the neighbouring rows' 64 bytes of the same K-tile."""
    *lead, rows, K = t.shape
    assert K % 32 == 0
    return t.reshape(*lead, rows, K // 32, 32).transpose(-3, -2).contiguous()


def unkblock(t: torch.Tensor) -> torch.Tensor:
    """Inverse of kblock: [.., K/32, rows, 32] -> [.., rows, K]."""
    *lead, kb, rows, _ = t.shape
    return t.transpose(-3, -2).reshape(*lead, rows, kb * 32).contiguous()


# ------------------------------------------------------------------------------------------ training layout
# The flat fp32 parameter / gradient buffers of the training engine (csrc/train_engine.hip) hold the same packed tensors as
# above (fp32; no K-blocked copies; the L skip-conv biases separately; the transformer's tensors layer-major).  An entry
# knows how to build its packed tensor from a state dict in the reference layout and how to write it back -- the latter is
# what turns the flat gradient buffer into per-parameter gradients under the reference's names, and the flat master buffer
# into a checkpoint the reference can load.
class _Entry:
    def __init__(self, name, shape, pack, unpack):
        self.name, self.shape, self.pack, self.unpack = name, tuple(shape), pack, unpack


def _unmat(p: torch.Tensor, n: int, k: int) -> torch.Tensor:
    return p[:n, :k].clone()


def _wavenet_entries(prefix: str, cin: int, cout: int, S: int, L: int) -> List[_Entry]:
    cinp, cp, cn = padk(cin), padk(cout), padn(cout)
    f32 = _lib.DN_F32
    blocks = [(s, i) for s in range(S) for i in range(L)]
    bk = lambda s, i: f"{prefix}stacks.{s}.blocks.{i}."
    last = lambda i: f"{prefix}stacks.{S - 1}.blocks.{i}.skip_conv."

    def conv_unpack(p, sd, key, c_in):  # [k, cn, Kp] -> [cout, cin, k]
        sd[key] = p[:, :cout, :c_in].permute(1, 2, 0).contiguous()

    def stack_unpack(p, sd, fmt, fn):
        for n, (s, i) in enumerate(blocks):
            fn(p[n], sd, fmt(s, i))

    return [
        _Entry(prefix + "init_W", (3, cn, cinp), lambda sd: _conv(sd[prefix + "init_conv.weight"], f32),
               lambda p, sd: conv_unpack(p, sd, prefix + "init_conv.weight", cin)),
        _Entry(prefix + "init_b", (cp,), lambda sd: _vec(sd[prefix + "init_conv.bias"], cp),
               lambda p, sd: sd.__setitem__(prefix + "init_conv.bias", p[:cout].clone())),
        _Entry(prefix + "conv_W", (S * L, 3, cn, cp), lambda sd: torch.stack([_conv(sd[bk(s, i) + "conv.weight"], f32) for s, i in blocks]),
               lambda p, sd: stack_unpack(p, sd, lambda s, i: bk(s, i) + "conv.weight", lambda q, d, k: conv_unpack(q, d, k, cout))),
        _Entry(prefix + "conv_b", (S * L, cp), lambda sd: torch.stack([_vec(sd[bk(s, i) + "conv.bias"], cp) for s, i in blocks]),
               lambda p, sd: stack_unpack(p, sd, lambda s, i: bk(s, i) + "conv.bias", lambda q, d, k: d.__setitem__(k, q[:cout].clone()))),
        _Entry(prefix + "res_W", (S * L, cn, cp), lambda sd: torch.stack([_mat(sd[bk(s, i) + "res_conv.weight"][:, :, 0], f32) for s, i in blocks]),
               lambda p, sd: stack_unpack(p, sd, lambda s, i: bk(s, i) + "res_conv.weight",
                                          lambda q, d, k: d.__setitem__(k, _unmat(q, cout, cout).unsqueeze(-1)))),
        _Entry(prefix + "res_b", (S * L, cp), lambda sd: torch.stack([_vec(sd[bk(s, i) + "res_conv.bias"], cp) for s, i in blocks]),
               lambda p, sd: stack_unpack(p, sd, lambda s, i: bk(s, i) + "res_conv.bias", lambda q, d, k: d.__setitem__(k, q[:cout].clone()))),
        _Entry(prefix + "skip_W", (L, cn, cp), lambda sd: torch.stack([_mat(sd[last(i) + "weight"][:, :, 0], f32) for i in range(L)]),
               lambda p, sd: [sd.__setitem__(last(i) + "weight", _unmat(p[i], cout, cout).unsqueeze(-1)) for i in range(L)]),
        _Entry(prefix + "skip_b", (L, cp), lambda sd: torch.stack([_vec(sd[last(i) + "bias"], cp) for i in range(L)]),
               lambda p, sd: [sd.__setitem__(last(i) + "bias", p[i, :cout].clone()) for i in range(L)]),
        _Entry(prefix + "final_W", (cn, cp), lambda sd: _mat(sd[prefix + "final_conv.weight"][:, :, 0], f32),
               lambda p, sd: sd.__setitem__(prefix + "final_conv.weight", _unmat(p, cout, cout).unsqueeze(-1))),
        _Entry(prefix + "final_b", (cp,), lambda sd: _vec(sd[prefix + "final_conv.bias"], cp),
               lambda p, sd: sd.__setitem__(prefix + "final_conv.bias", p[:cout].clone())),
    ]


def _tf_layer_entries(prefix: str, l: int, dim: int, heads: int, dim_head: int) -> List[_Entry]:
    inner = int(dim * 4 * 2 / 3)
    ip, Dp, Dn, hd, in_n = padk(inner), padk(dim), padn(dim), heads * dim_head, padn(inner)
    f32 = _lib.DN_F32
    p_ = f"{prefix}layers.{l}."
    rows = _geglu_rows(inner)
    keep = rows >= 0

    def ffin_pack(sd):
        w = sd[p_ + "5.0.weight"].float()
        out = torch.zeros(2 * ip, Dp)
        out[keep, :dim] = w[rows[keep]]
        return out

    def ffin_unpack(p, sd):
        w = torch.zeros(2 * inner, dim)
        w[rows[keep]] = p[keep, :dim]
        sd[p_ + "5.0.weight"] = w

    def ffin_b_pack(sd):
        out = torch.zeros(2 * ip)
        out[keep] = sd[p_ + "5.0.bias"].float()[rows[keep]]
        return out

    def ffin_b_unpack(p, sd):
        b = torch.zeros(2 * inner)
        b[rows[keep]] = p[keep]
        sd[p_ + "5.0.bias"] = b

    def qkv_unpack(p, sd):
        sd[p_ + "1.to_q.weight"] = p[:hd, :dim].clone()
        sd[p_ + "1.to_kv.weight"] = p[hd:3 * hd, :dim].clone()

    return [
        _Entry(p_ + "qkv_W", (padn(3 * hd), Dp), lambda sd: _mat(torch.cat([sd[p_ + "1.to_q.weight"], sd[p_ + "1.to_kv.weight"]], dim=0), f32),
               qkv_unpack),
        _Entry(p_ + "out_W", (Dn, hd), lambda sd: _mat(sd[p_ + "1.to_out.weight"], f32),
               lambda p, sd: sd.__setitem__(p_ + "1.to_out.weight", _unmat(p, dim, hd))),
        _Entry(p_ + "ffin_W", (2 * ip, Dp), ffin_pack, ffin_unpack),
        _Entry(p_ + "ffin_b", (2 * ip,), ffin_b_pack, ffin_b_unpack),
        _Entry(p_ + "ffconv_W", (3, in_n, ip), lambda sd: _conv(sd[p_ + "5.2.1.weight"], f32),
               lambda p, sd: sd.__setitem__(p_ + "5.2.1.weight", p[:, :inner, :inner].permute(1, 2, 0).contiguous())),
        _Entry(p_ + "ffconv_b", (ip,), lambda sd: _vec(sd[p_ + "5.2.1.bias"], ip),
               lambda p, sd: sd.__setitem__(p_ + "5.2.1.bias", p[:inner].clone())),
        _Entry(p_ + "ffout_W", (Dn, ip), lambda sd: _mat(sd[p_ + "5.3.weight"], f32),
               lambda p, sd: sd.__setitem__(p_ + "5.3.weight", _unmat(p, dim, inner))),
        _Entry(p_ + "ffout_b", (Dp,), lambda sd: _vec(sd[p_ + "5.3.bias"], Dp),
               lambda p, sd: sd.__setitem__(p_ + "5.3.bias", p[:dim].clone())),
        _Entry(p_ + "g1", (dim,), lambda sd: sd[p_ + "0.gamma"].float().clone(), lambda p, sd: sd.__setitem__(p_ + "0.gamma", p.clone())),
        _Entry(p_ + "g2", (dim,), lambda sd: sd[p_ + "4.gamma"].float().clone(), lambda p, sd: sd.__setitem__(p_ + "4.gamma", p.clone())),
    ]


def vae_train_entries(dim: int, mults: List[int], depth: int, heads: int, dim_head: int, stacks: int, layers: int,
                      vocab: int) -> List[_Entry]:
    """Table of the VAE training engine's packed tensors, in the order of dn_vae_train_offsets."""
    f32 = _lib.DN_F32
    Dp, Dn, Vn = padk(dim), padn(dim), padn(vocab)
    ents: List[_Entry] = []
    cur = dim
    for n, m in enumerate(mults):
        ents += _wavenet_entries(f"encoder_wave.{n}.", cur, cur // m, stacks, layers)
        cur //= m
    first = True
    for n, m in enumerate(reversed(mults)):
        tgt = cur * m
        cin = cur // 2 if first else cur
        first = False
        ents += _wavenet_entries(f"decoder_wave.{n}.", cin, tgt, stacks, layers)
        cur = tgt
    for l in range(depth):
        ents += _tf_layer_entries("decoder_tf.", l, dim, heads, dim_head)
    ents += [
        _Entry("decoder_tf.pred_gamma", (dim,), lambda sd: sd["decoder_tf.to_pred.0.gamma"].float().clone(),
               lambda p, sd: sd.__setitem__("decoder_tf.to_pred.0.gamma", p.clone())),
        _Entry("decoder_tf.pred_W", (Dn, Dp), lambda sd: _mat(sd["decoder_tf.to_pred.1.weight"], f32),
               lambda p, sd: sd.__setitem__("decoder_tf.to_pred.1.weight", _unmat(p, dim, dim))),
        _Entry("decoder_lm.W", (Vn, Dp), lambda sd: _mat(sd["decoder_lm.weight"], f32),
               lambda p, sd: sd.__setitem__("decoder_lm.weight", _unmat(p, vocab, dim))),
        _Entry("decoder_lm.b", (Vn,), lambda sd: _vec(sd["decoder_lm.bias"], Vn),
               lambda p, sd: sd.__setitem__("decoder_lm.bias", p[:vocab].clone())),
    ]
    return ents


def pack_flat(sd: SD, entries: List[_Entry], offsets: List[int], total: int) -> torch.Tensor:
    """State dict (reference layout) -> flat fp32 buffer of `total` elements."""
    flat = torch.zeros(total, dtype=torch.float32)
    for e, off in zip(entries, offsets):
        t = e.pack(sd).float()
        assert tuple(t.shape) == e.shape, (e.name, tuple(t.shape), e.shape)
        flat[off: off + t.numel()] = t.reshape(-1)
    return flat


def unpack_flat(flat: torch.Tensor, entries: List[_Entry], offsets: List[int]) -> SD:
    """Flat buffer (parameters or gradients) -> tensors under the reference's state-dict names and shapes."""
    flat = flat.detach().float().cpu()
    sd: SD = {}
    for e, off in zip(entries, offsets):
        n = 1
        for s in e.shape:
            n *= s
        e.unpack(flat[off: off + n].view(e.shape), sd)
    return sd


def eps_train_entries(cfg) -> List[_Entry]:
    """Table of the diffusion training engine's packed tensors, in the order of dn_eps_train_offsets (fp32; the conditioning
    path first: to_time_cond, then the 2 S L FiLM and 2 depth adaptive-norm projections stacked as [gamma (Dp) ; beta (Dp)] rows)."""
    f32 = _lib.DN_F32
    D, Dp, Dn, zl = cfg.dim, padk(cfg.dim), padn(cfg.dim), cfg.latent_dim
    zp, C = padk(zl), cfg.dim * cfg.dim_cond_mult
    mods = [f"wavenet.stacks.{s}.blocks.{i}.to_time_cond." for s in range(cfg.wavenet_stacks) for i in range(cfg.wavenet_layers)]
    mods += [f"transformer.layers.{l}.{j}.to_gamma_beta." for l in range(cfg.depth) for j in (0, 4)]
    n_cond = len(mods) * 2 * Dp

    def cond_pack(sd):
        out = torch.zeros(padn(n_cond), C)
        for n, key in enumerate(mods):
            w = sd[key + "weight"].float()  # [2D, C]: [gamma ; beta]
            out[n * 2 * Dp: n * 2 * Dp + D] = w[:D]
            out[n * 2 * Dp + Dp: n * 2 * Dp + Dp + D] = w[D:]
        return out

    def cond_unpack(p, sd):
        for n, key in enumerate(mods):
            sd[key + "weight"] = torch.cat([p[n * 2 * Dp: n * 2 * Dp + D], p[n * 2 * Dp + Dp: n * 2 * Dp + Dp + D]], dim=0).clone()

    def condb_pack(sd):
        out = torch.zeros(n_cond)
        for n, key in enumerate(mods):
            b = sd[key + "bias"].float()
            out[n * 2 * Dp: n * 2 * Dp + D] = b[:D]
            out[n * 2 * Dp + Dp: n * 2 * Dp + Dp + D] = b[D:]
        return out

    def condb_unpack(p, sd):
        for n, key in enumerate(mods):
            sd[key + "bias"] = torch.cat([p[n * 2 * Dp: n * 2 * Dp + D], p[n * 2 * Dp + Dp: n * 2 * Dp + Dp + D]]).clone()

    ents = [
        _Entry("w_freq", (D // 2,), lambda sd: sd["to_time_cond.0.weights"].float().clone(),
               lambda p, sd: sd.__setitem__("to_time_cond.0.weights", p.clone())),
        _Entry("tc_W", (C, D + 1), lambda sd: sd["to_time_cond.1.weight"].float().clone(),
               lambda p, sd: sd.__setitem__("to_time_cond.1.weight", p.clone())),
        _Entry("tc_b", (C,), lambda sd: sd["to_time_cond.1.bias"].float().clone(), lambda p, sd: sd.__setitem__("to_time_cond.1.bias", p.clone())),
        _Entry("cond_W", (padn(n_cond), C), cond_pack, cond_unpack),
        _Entry("cond_b", (n_cond,), condb_pack, condb_unpack),
        _Entry("init_W", (Dn, zp), lambda sd: _mat(sd["init_conv.weight"][:, :, 0], f32),
               lambda p, sd: sd.__setitem__("init_conv.weight", _unmat(p, D, zl).unsqueeze(-1))),
        _Entry("init_b", (Dp,), lambda sd: _vec(sd["init_conv.bias"], Dp), lambda p, sd: sd.__setitem__("init_conv.bias", p[:D].clone())),
    ]
    ents += _wavenet_entries("wavenet.", D, D, cfg.wavenet_stacks, cfg.wavenet_layers)
    for l in range(cfg.depth):
        ents += _tf_layer_entries("transformer.", l, D, cfg.heads, cfg.dim_head)[:8]  # no learned gammas in conditional norms
    ents += [
        _Entry("pred_gamma", (D,), lambda sd: sd["transformer.to_pred.0.gamma"].float().clone(),
               lambda p, sd: sd.__setitem__("transformer.to_pred.0.gamma", p.clone())),
        _Entry("pred_W", (Dn, Dp), lambda sd: _mat(sd["transformer.to_pred.1.weight"], f32),
               lambda p, sd: sd.__setitem__("transformer.to_pred.1.weight", _unmat(p, D, D))),
        _Entry("final_W", (padn(zl), Dp), lambda sd: _mat(sd["final_proj.weight"], f32),
               lambda p, sd: sd.__setitem__("final_proj.weight", _unmat(p, zl, D))),
        _Entry("final_b", (zp,), lambda sd: _vec(sd["final_proj.bias"], zp), lambda p, sd: sd.__setitem__("final_proj.bias", p[:zl].clone())),
    ]
    return ents
