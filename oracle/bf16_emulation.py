"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (diffnorm_amd/).

A CPU model of the HIP engines' DN_BF16 arithmetic: the same functions as oracle/diffnorm_oracle.py (each citing the same
reference lines), with a bf16 rounding exactly where the engines store or feed a tensor as bf16 and fp32 everywhere else
(contraction operands -- weights and activations -- are bf16, accumulation is fp32, the transformer's residual
stream, the conditioning path, norm statistics and softmax stay fp32).  It exists so that the engine-level bf16 tests have a
TIGHT bound: engine-bf16 vs this model differs only by fp32 summation order, the fast transcendentals and the rare bf16
rounding flip those cause, whereas engine-bf16 vs the fp32 golden outputs also contains the (much larger) effect of the
operand rounding itself.  `skip` names rounding points to leave in fp32 (ablations: which rounding costs what).

Rounding points (engine source: csrc/engine.hip run_wavenet / run_transformer / eps_core / dn_vae_*):
  in      activations converted to bf16 at the entry (dn_convert_rows)
  w       every contraction weight
  wn      WaveNet hidden states: init conv, res conv, gate outputs, skip sum
  xn      the norm's output row * gamma (split RMSNorm producer), before the 1/|x| factor
  rb      beta of the adaptive norms as operand of the beta . W^T contraction
  qkv, p (softmax probabilities fed to P.V and to the denominator), ao, gg (GEGLU output), fc (FFN conv output), tp (to_pred)
"""
import math
from typing import Optional, Set

import torch
import torch.nn.functional as F

import diffnorm_oracle as O
from diffnorm_oracle import sub

Tensor = torch.Tensor


class Rounder:
    """kind "bf16" (8 mantissa bits, fp32's exponent range) or "f16" (11 mantissa bits, |x| <= 65504, subnormals below 2^-14:
    the operand format of v_mfma_f32_16x16x32_f16, which issues at the bf16 rate on gfx950).  For "f16" the rounder counts,
    per rounding point, the values that overflow to inf and the non-zero values that fall into the subnormal range (they keep
    fewer than 11 bits) or flush to zero -- the two ways the narrower exponent can hurt where bf16 cannot."""

    F16_MAX, F16_MIN_NORMAL, F16_MIN_SUB = 65504.0, 2.0 ** -14, 2.0 ** -24

    def __init__(self, skip: Optional[Set[str]] = None, kind: str = "bf16"):
        assert kind in ("bf16", "f16")
        self.skip, self.kind = set(skip or ()), kind
        self.stats = {}  # point -> [elements, overflow, subnormal, flushed to zero, max |x|]

    def __call__(self, t: Tensor, point: str) -> Tensor:
        if point in self.skip:
            return t
        if self.kind == "bf16":
            return t.to(torch.bfloat16).float()
        a = t.detach().abs()
        s = self.stats.setdefault(point, [0, 0, 0, 0, 0.0])
        s[0] += a.numel()
        s[1] += int((a > self.F16_MAX).sum())
        s[2] += int(((a < self.F16_MIN_NORMAL) & (a >= self.F16_MIN_SUB)).sum())
        s[3] += int(((a < self.F16_MIN_SUB / 2) & (a > 0)).sum())
        s[4] = max(s[4], float(a.max()) if a.numel() else 0.0)
        return t.to(torch.float16).float()


def _conv(x: Tensor, w: Tensor, b: Optional[Tensor], dil: int, R: Rounder) -> Tensor:
    """causal conv on already-rounded activations; fp32 accumulation, fp32 bias added afterwards (the epilogue)."""
    y = O.causal_conv1d(x, R(w, "w"), None, dil)
    return y if b is None else y + b


def wavenet(sd, x_b: Tensor, stacks: int, layers: int, t: Optional[Tensor], R: Rounder) -> Tensor:
    """run_wavenet: returns the final 1x1 conv's fp32 accumulators + bias (the caller applies its destination's rounding)."""
    h = R(_conv(x_b, sd["init_conv.weight"], sd["init_conv.bias"], 1, R), "wn")
    inputs = [h] * layers
    for s in range(stacks):
        outs = []
        for i in range(layers):
            p = f"stacks.{s}.blocks.{i}."
            res = R(_conv(inputs[i], sd[p + "res_conv.weight"], sd[p + "res_conv.bias"], 1, R), "wn")
            hh = _conv(inputs[i], sd[p + "conv.weight"], sd[p + "conv.bias"], 2 ** i, R)
            if p + "to_time_cond.weight" in sd:  # FiLM in fp32 on the accumulators (:517-527)
                g, b = F.linear(t, sd[p + "to_time_cond.weight"], sd[p + "to_time_cond.bias"]).chunk(2, dim=-1)
                hh = hh * g.unsqueeze(1) + b.unsqueeze(1)
            outs.append(R(hh.tanh() * hh.sigmoid() + res, "wn"))
        inputs = outs
    last = f"stacks.{stacks - 1}.blocks."
    total = sum(_conv(inputs[i], sd[f"{last}{i}.skip_conv.weight"], None, 1, R) for i in range(layers))
    total = R(total + sum(sd[f"{last}{i}.skip_conv.bias"] for i in range(layers)), "wn")
    return _conv(total, sd["final_conv.weight"], sd["final_conv.bias"], 1, R)


def _attention(q: Tensor, k: Tensor, v: Tensor, mask: Optional[Tensor], heads: int, R: Rounder) -> Tensor:
    """attn_kernel: scores and softmax in fp32, P rounded to bf16 for both P.V and the denominator."""
    B, T, hd = q.shape
    dh = hd // heads
    q, k, v = (z.view(B, T, heads, dh).transpose(1, 2) for z in (q, k, v))
    sim = torch.matmul(q, k.transpose(-1, -2)) * (dh ** -0.5)
    if mask is not None:
        sim = sim.masked_fill(~mask.view(B, 1, 1, T), -torch.finfo(sim.dtype).max)
    p = R(torch.exp(sim - sim.amax(dim=-1, keepdim=True)), "p")
    out = torch.matmul(p, v) / p.sum(dim=-1, keepdim=True)
    return out.transpose(1, 2).reshape(B, T, hd)


def transformer(sd, x: Tensor, depth: int, heads: int, mask: Optional[Tensor], t: Optional[Tensor], R: Rounder, first_norm_split: bool,
                pred_round: bool) -> Tensor:
    """run_transformer with the split RMSNorm: a norm's producer stores bf16(x * gamma), its consumer scales the fp32 accumulators
    by sqrt(D) / |x| and adds beta . W^T.  x: fp32 residual stream.  first_norm_split: layer 0's attention norm came out of the
    contraction that opened the stream (eps-predictor); otherwise it is the stand-alone kernel (full norm, then rounded: VAE)."""
    D = x.shape[-1]

    def gb(key):
        if t is None:
            return sd[key + "gamma"], None
        g, b = F.linear(t, sd[key + "to_gamma_beta.weight"], sd[key + "to_gamma_beta.bias"]).chunk(2, dim=-1)
        return g.unsqueeze(1), b.unsqueeze(1)

    def produce(xr, key):
        g, b = gb(key) if key != "to_pred.0." else (sd["to_pred.0.gamma"], None)
        scale = (D ** 0.5) / xr.norm(dim=-1, keepdim=True).clamp(min=1e-12)
        return R(xr * g, "xn"), scale, b

    def consume(xn, scale, beta, w, bias):
        acc = F.linear(xn, R(w, "w")) * scale
        if beta is not None:
            acc = acc + F.linear(R(beta, "rb"), R(w, "w"))
        return acc if bias is None else acc + bias

    for layer in range(depth):
        p = f"layers.{layer}."
        if layer == 0 and not first_norm_split:
            g, b = gb(p + "0.")
            full = F.normalize(x, dim=-1) * (D ** 0.5) * g
            xn, scale, beta = R(full if b is None else full + b, "xn"), 1.0, None
        else:
            xn, scale, beta = produce(x, p + "0.")
        a = sub(sd, p + "1.")
        qkv = R(consume(xn, scale, beta, torch.cat([a["to_q.weight"], a["to_kv.weight"]], dim=0), None), "qkv")
        q, k, v = qkv.chunk(3, dim=-1)
        ao = R(_attention(q, k, v, mask, heads, R), "ao")
        x = x + F.linear(ao, R(a["to_out.weight"], "w"))
        xn, scale, beta = produce(x, p + "4.")
        f = sub(sd, p + "5.")
        h = consume(xn, scale, beta, f["0.weight"], f["0.bias"])
        val, gate = h.chunk(2, dim=-1)
        gg = R(F.gelu(gate) * val, "gg")
        fc = R(_conv(gg, f["2.1.weight"], f["2.1.bias"], 1, R), "fc")
        x = x + F.linear(fc, R(f["3.weight"], "w"), f["3.bias"])
    xn, scale, _ = produce(x, "to_pred.0.")
    out = F.linear(xn, R(sd["to_pred.1.weight"], "w")) * scale
    return R(out, "tp") if pred_round else out


def eps_forward(sd, cfg, x: Tensor, times: Tensor, mask: Tensor, skip: Optional[Set[str]] = None, R: Optional[Rounder] = None) -> Tensor:
    """dn_eps_forward in DN_BF16 / DN_F16 mode (Model.forward latent_module.py:828-876)."""
    R = R or Rounder(skip)
    t = O.time_cond(sd, times)  # conditioning path: fp32 throughout
    h = R(_conv(R(x, "in"), sd["init_conv.weight"], sd["init_conv.bias"], 1, R), "wn")
    h = wavenet(sub(sd, "wavenet."), h, cfg.wavenet_stacks, cfg.wavenet_layers, t, R)
    h = h + O.positional_embedding(mask, cfg.dim)  # POSEMB epilogue, fp32 residual stream
    h = transformer(sub(sd, "transformer."), h, cfg.depth, cfg.heads, mask, t, R, first_norm_split=True, pred_round=True)
    return F.linear(h, R(sd["final_proj.weight"], "w"), sd["final_proj.bias"])


def vae_encode_params(sd, cfg, feat: Tensor, skip: Optional[Set[str]] = None, R: Optional[Rounder] = None) -> Tensor:
    """dn_vae_encode_params (:1099-1106): posterior parameters, fp32 out of the last WaveNet."""
    R = R or Rounder(skip)
    x = R(feat, "in")
    n = len(cfg.chan_mults())
    for i in range(n):
        y = wavenet(sub(sd, f"encoder_wave.{i}."), x, cfg.stacks, cfg.layers, None, R)
        x = y if i == n - 1 else R(y, "wn")
    return x


def vae_decode(sd, cfg, latent: Tensor, mask: Tensor, skip: Optional[Set[str]] = None, R: Optional[Rounder] = None):
    """dn_vae_decode (:1109-1116) -> (recon fp32, logits fp32)."""
    R = R or Rounder(skip)
    x = R(latent, "in")
    n = len(cfg.chan_mults())
    for i in range(n):
        y = wavenet(sub(sd, f"decoder_wave.{i}."), x, cfg.stacks, cfg.layers, None, R)
        x = y if i == n - 1 else R(y, "wn")
    dec = transformer(sub(sd, "decoder_tf."), x, cfg.depth, cfg.heads, mask, None, R, first_norm_split=False, pred_round=False)
    logits = F.linear(R(dec, "in"), R(sd["decoder_lm.weight"], "w"), sd["decoder_lm.bias"])
    return dec, logits
