"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the DiffNorm latent-diffusion hot path.

A from-scratch, functional restatement (plain PyTorch fp32 on CPU, NumPy float64 for
the schedule tables) of the reference algorithm, every function citing the reference
``file:line`` it follows.  It consumes tensors in the reference's *state-dict layout*
 so the same weights feed the oracle and the HIP packer.

Rules (tier section 3): only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this file, and only as the checker / reported baseline.
The product path (``diffnorm_amd/``) never imports it and fails loudly when the HIP
library is missing.

Parity pin: this oracle is checked against outputs of the real reference, imported in
the build container by ``oracle/ref_loader.py``; the resulting vectors are committed
under ``tests/golden/`` with their generator ``oracle/gen_golden.py`` and verified by
``tests/test_oracle_golden.py`` (CPU, no reference needed at test time).

Layout convention: activations are ``[B, T, C]`` (channels-last) at every function
boundary; the reference's ``[B, C, T]`` ping-pong is internal to ``causal_conv1d``.
Paths below are relative to /root/reference/fairseq/models/text_to_speech/ unless
they start with ``fairseq/``.
"""
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------- configs
@dataclass
class EpsConfig:
    """Hyper-parameters of the eps-predictor ``Model`` (latent_module.py:709-728)."""

    dim: int = 512
    latent_dim: int = 128
    depth: int = 12
    heads: int = 8
    dim_head: int = 64
    wavenet_layers: int = 8
    wavenet_stacks: int = 4
    dim_cond_mult: int = 4
    # conditional variant (use_cond=True; latent_module.py:709-728, 752-773): 0 = unconditional
    dim_prompt: int = 0
    num_latents_m: int = 64
    resampler_depth: int = 2


@dataclass
class VaeConfig:
    """``SpeechVAEEncoderDecoder`` hyper-parameters (latent_module.py:1035-1096)."""

    dim: int = 768
    latent_dim: int = 128
    depth: int = 6
    heads: int = 8
    dim_head: int = 96
    stacks: int = 2
    layers: int = 3
    vocab: int = 1004

    def chan_mults(self) -> List[int]:
        # latent_module.py:1044-1051 (any other latent_dim is a NameError upstream);
        # ``latent_dim`` is the upstream constructor flag, the actual width is ``z``
        return {16: [4, 3, 2], 32: [4, 3], 128: [3]}[self.latent_dim]

    @property
    def z(self) -> int:
        """Actual latent width: dim / prod(mults) / 2 (== latent_dim when dim == 768)."""
        c = self.dim
        for m in self.chan_mults():
            c //= m
        return c // 2


def sub(sd: SD, prefix: str) -> SD:
    """View of a state dict below ``prefix`` (prefix stripped)."""
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


# --------------------------------------------------------------------------- helpers
def lengths_to_mask(lengths: Tensor, max_len: Optional[int] = None) -> Tensor:
    """fairseq/data/data_utils.py:542-552 -- ``arange < len``."""
    max_len = int(lengths.max()) if max_len is None else max_len
    return torch.arange(max_len).view(1, -1) < lengths.view(-1, 1)


def label_smoothed_nll_loss(lprobs: Tensor, target: Tensor, epsilon: float, ignore_index: int):
    """fairseq/criterions/label_smoothed_cross_entropy.py:34-51 (reduce=True)."""
    tgt = target.view(-1, 1)
    keep = tgt.ne(ignore_index)
    nll = (-lprobs.gather(-1, tgt) * keep).sum()
    smooth = (-lprobs.sum(-1, keepdim=True) * keep).sum()
    eps_i = epsilon / (lprobs.size(-1) - 1)
    return (1.0 - epsilon - eps_i) * nll + eps_i * smooth, nll


def causal_conv1d(x: Tensor, w: Tensor, b: Optional[Tensor], dilation: int = 1) -> Tensor:
    """CausalConv1d.forward latent_module.py:476-488 on channels-last input.

    x [B,T,Cin], w [Cout,Cin,k]: left zero-pad dilation*(k-1), stride 1, no right pad.
    """
    k = w.shape[-1]
    xc = F.pad(x.transpose(1, 2), (dilation * (k - 1), 0))
    return F.conv1d(xc, w, b, dilation=dilation).transpose(1, 2)


def sinusoidal_table(num: int, dim: int) -> Tensor:
    """fairseq/modules/sinusoidal_positional_embedding.py:36-58 with padding_idx=0.

    rows [sin(p f_j) | cos(p f_j)], f_j = exp(-j ln(1e4)/(dim/2-1)); row 0 zeroed.
    """
    half = dim // 2
    f = torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1)))
    ang = torch.arange(num, dtype=torch.float).unsqueeze(1) * f.unsqueeze(0)
    tab = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)
    if dim % 2 == 1:
        tab = torch.cat([tab, torch.zeros(num, 1)], dim=1)
    tab[0] = 0
    return tab


def positional_embedding(mask: Tensor, dim: int) -> Tensor:
    """PositionalEmbedding(1024, dim, 0) applied to a bool mask, latent_module.py:774-779,867.

    positions = cumsum(mask)*mask (fairseq/utils.py:256-266): valid frames 1..len, pads 0.
    """
    m = mask.int()
    pos = (torch.cumsum(m, dim=1) * m).long()
    tab = sinusoidal_table(max(1024, mask.shape[1] + 1), dim)
    return tab[pos]


# --------------------------------------------------------------------------- blocks
def time_cond(sd: SD, times: Tensor) -> Tensor:
    """LearnedSinusoidalPosEmb + Linear + SiLU: latent_module.py:104-116, 741-745.

    times int64 [B] (raw step index, not normalised) -> [B, dim*dim_cond_mult].
    """
    x = times.view(-1, 1)
    freqs = x * sd["to_time_cond.0.weights"].view(1, -1) * 2 * math.pi
    four = torch.cat((freqs.sin(), freqs.cos()), dim=-1)
    four = torch.cat((x, four), dim=-1)
    return F.silu(F.linear(four, sd["to_time_cond.1.weight"], sd["to_time_cond.1.bias"]))


def wavenet_block(sd: SD, x: Tensor, dilation: int, t: Optional[Tensor]):
    """WavenetResBlock.forward latent_module.py:513-536.  Returns (residual_out, skip|None)."""
    res = causal_conv1d(x, sd["res_conv.weight"], sd["res_conv.bias"])
    h = causal_conv1d(x, sd["conv.weight"], sd["conv.bias"], dilation)
    if "to_time_cond.weight" in sd:
        gb = F.linear(t, sd["to_time_cond.weight"], sd["to_time_cond.bias"])
        g, b = gb.chunk(2, dim=-1)  # [gamma ; beta] :517-519
        h = h * g.unsqueeze(1) + b.unsqueeze(1)
    h = h.tanh() * h.sigmoid()  # same tensor on both sides, :528
    h = h + res
    skip = None
    if "skip_conv.weight" in sd:
        skip = causal_conv1d(h, sd["skip_conv.weight"], sd["skip_conv.bias"])
    return h, skip


def wavenet(sd: SD, x: Tensor, stacks: int, layers: int, t: Optional[Tensor] = None) -> Tensor:
    """Wavenet / WavenetEncoder forward: latent_module.py:566-582, 613-617, 1028-1032.

    Stack 0 feeds one tensor to all blocks; stack s>0 feeds block i with stack s-1's
    block-i output; only the last stack has skips, summed, then final 1x1 conv.
    """
    h = causal_conv1d(x, sd["init_conv.weight"], sd["init_conv.bias"])
    inputs = [h] * layers
    skips = None
    for s in range(stacks):
        outs, sk = [], []
        for i in range(layers):
            o, k = wavenet_block(sub(sd, f"stacks.{s}.blocks.{i}."), inputs[i], 2 ** i, t)
            outs.append(o)
            sk.append(k)
        inputs = outs
        if s == stacks - 1:
            skips = sk
    total = torch.stack(skips).sum(dim=0)
    return causal_conv1d(total, sd["final_conv.weight"], sd["final_conv.bias"])


def rms_norm(x: Tensor, gamma: Optional[Tensor] = None, cond_wb=None, cond: Optional[Tensor] = None):
    """RMSNorm.forward latent_module.py:620-639: x/max(|x|,1e-12)*sqrt(D)*gamma [*g_c + b_c]."""
    out = F.normalize(x, dim=-1) * (x.shape[-1] ** 0.5)
    if gamma is not None:
        out = out * gamma
    if cond_wb is not None:
        g, b = F.linear(cond, cond_wb[0], cond_wb[1]).chunk(2, dim=-1)  # [gamma ; beta] :637
        out = out * g.unsqueeze(1) + b.unsqueeze(1)
    return out


_ATTN_DROP = None  # train mode: dict(p, which, keep) set by `attention_dropout`


class attention_dropout:
    """Context manager: train mode of the transformer named `which` ("vae" decoder or "eps" predictor).  The reference applies
    nn.Dropout(0.1) to the attention probabilities (latent_module.py:338, 668) with torch's generator; a restatement cannot
    re-draw that stream, so the keep mask is supplied: keep(layer, B, heads, T, Tk) -> bool [B, heads, T, Tk]."""

    def __init__(self, which: str, p: float, keep):
        self.cfg = dict(which=which, p=float(p), keep=keep)

    def __enter__(self):
        global _ATTN_DROP
        self.prev, _ATTN_DROP = _ATTN_DROP, self.cfg

    def __exit__(self, *exc):
        global _ATTN_DROP
        _ATTN_DROP = self.prev


def attention(sd: SD, x: Tensor, mask: Optional[Tensor], heads: int, context: Optional[Tensor] = None, drop=None) -> Tensor:
    """Attention.forward + Attend.forward (non-flash): latent_module.py:934-950, 299-343.

    Only keys are masked (fill -finfo.max); no biases; eval mode unless `drop` = (which, layer) names a transformer that an
    enclosing `attention_dropout` put into train mode (:338).  `context` (cross-attention, :935-943): keys and values come from
    it, `mask` then masks ITS positions.
    """
    B, T, _ = x.shape
    ctx = x if context is None else context
    Tk = ctx.shape[1]
    q = F.linear(x, sd["to_q.weight"])
    k, v = F.linear(ctx, sd["to_kv.weight"]).chunk(2, dim=-1)  # [k ; v] :945
    dh = q.shape[-1] // heads
    q = q.view(B, T, heads, dh).transpose(1, 2)  # (h d), h major
    k, v = (z.view(B, Tk, heads, dh).transpose(1, 2) for z in (k, v))
    sim = torch.matmul(q, k.transpose(-1, -2)) * (dh ** -0.5)
    if mask is not None:
        sim = sim.masked_fill(~mask.view(B, 1, 1, Tk), -torch.finfo(sim.dtype).max)
    attn = sim.softmax(dim=-1)
    if drop is not None and _ATTN_DROP is not None and _ATTN_DROP["which"] == drop[0]:
        keep = _ATTN_DROP["keep"](drop[1], B, heads, T, Tk)
        attn = attn * keep.to(attn.dtype) / (1.0 - _ATTN_DROP["p"])
    out = torch.matmul(attn, v)
    out = out.transpose(1, 2).reshape(B, T, heads * dh)
    return F.linear(out, sd["to_out.weight"])


def feed_forward(sd: SD, x: Tensor) -> Tensor:
    """FeedForward + GEGLU: latent_module.py:881-903.

    Linear(D->2*inner) ; gelu_exact(gate)*value (first half = value) ; causal conv k=3 ;
    Linear(inner->D).
    """
    h = F.linear(x, sd["0.weight"], sd["0.bias"])
    val, gate = h.chunk(2, dim=-1)
    h = F.gelu(gate) * val
    h = causal_conv1d(h, sd["2.1.weight"], sd["2.1.bias"])
    return F.linear(h, sd["3.weight"], sd["3.bias"])


def transformer(sd: SD, x: Tensor, depth: int, heads: int, mask: Optional[Tensor], t: Optional[Tensor], context: Optional[Tensor] = None,
                which: Optional[str] = None):
    """ConditionableTransformer.forward latent_module.py:681-706; with `context` the layers carry the cross-attention block
    (:694-700: norm, attend to the resampled prompt latents without a mask, residual)."""
    for layer in range(depth):
        p = f"layers.{layer}."
        if t is not None:
            n1 = rms_norm(x, None, (sd[p + "0.to_gamma_beta.weight"], sd[p + "0.to_gamma_beta.bias"]), t)
        else:
            n1 = rms_norm(x, sd[p + "0.gamma"])
        x = attention(sub(sd, p + "1."), n1, mask, heads, drop=None if which is None else (which, layer)) + x
        if context is not None:
            nc = rms_norm(x, None, (sd[p + "2.to_gamma_beta.weight"], sd[p + "2.to_gamma_beta.bias"]), t)
            x = attention(sub(sd, p + "3."), nc, None, heads, context=context) + x
        if t is not None:
            n2 = rms_norm(x, None, (sd[p + "4.to_gamma_beta.weight"], sd[p + "4.to_gamma_beta.bias"]), t)
        else:
            n2 = rms_norm(x, sd[p + "4.gamma"])
        x = feed_forward(sub(sd, p + "5."), n2) + x
    x = rms_norm(x, sd["to_pred.0.gamma"])
    return F.linear(x, sd["to_pred.1.weight"])


# --------------------------------------------------------------------------- eps-predictor
def eps_forward(sd: SD, cfg: EpsConfig, x: Tensor, times: Tensor, mask: Tensor) -> Tensor:
    """Model.forward latent_module.py:828-876 (no prompt branch).

    x [B,T,latent] fp32, times [B] int64, mask [B,T] bool -> eps_hat [B,T,latent].
    """
    t = time_cond(sd, times)
    h = causal_conv1d(x, sd["init_conv.weight"], sd["init_conv.bias"])  # 1x1, :734,864
    h = wavenet(sub(sd, "wavenet."), h, cfg.wavenet_stacks, cfg.wavenet_layers, t)
    h = h + positional_embedding(mask, cfg.dim)
    h = transformer(sub(sd, "transformer."), h, cfg.depth, cfg.heads, mask, t, which="eps")
    return F.linear(h, sd["final_proj.weight"], sd["final_proj.bias"])


def perceiver_resampler(sd: SD, prompt: Tensor, prompt_mask: Tensor, heads: int) -> Tensor:
    """PerceiverResampler.forward latent_module.py:416-471: learned latents (+ sinusoidal positions 1..m) attend to
    [latents ; projected prompt] (cross_attn_include_queries, :935-943: the mask gets ones for the latents), feed-forward without
    the causal conv, residuals, final learned-gamma RMSNorm."""
    B = prompt.shape[0]
    x = F.linear(prompt, sd["proj_context.weight"], sd["proj_context.bias"])
    lat = sd["latents"]
    m, D = lat.shape
    lat = lat.unsqueeze(0).expand(B, -1, -1) + positional_embedding(torch.ones(B, m, dtype=torch.bool), D)
    heads_dim = sd["layers.0.0.to_q.weight"].shape[0]
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("layers."))
    for l in range(depth):
        a = sub(sd, f"layers.{l}.0.")
        ctx = torch.cat([lat, x], dim=1)
        cmask = torch.cat([torch.ones(B, m, dtype=torch.bool), prompt_mask], dim=1)
        lat = attention(a, lat, cmask, heads, context=ctx) + lat
        f = sub(sd, f"layers.{l}.1.")
        h = F.linear(lat, f["0.weight"], f["0.bias"])
        val, gate = h.chunk(2, dim=-1)
        lat = F.linear(F.gelu(gate) * val, f["2.weight"], f["2.bias"]) + lat
    return rms_norm(lat, sd["norm.gamma"])


def eps_forward_cond(sd: SD, cfg: EpsConfig, x: Tensor, times: Tensor, mask: Tensor, prompt: Tensor, prompt_mask: Tensor,
                     drop: Tensor) -> Tensor:
    """Model.forward with condition_on_prompt (latent_module.py:828-876).  drop [B] bool = the classifier-free-guidance drop mask
    (prob_mask_like, :843): dropped samples use null_prompt_cond / null_prompt_tokens."""
    t = time_cond(sd, times)
    masked = prompt.masked_fill(~prompt_mask.unsqueeze(2), 0.0)
    pc = F.silu(F.linear(masked.mean(dim=1), sd["to_prompt_cond.1.weight"], sd["to_prompt_cond.1.bias"]))  # mean over ALL positions
    pc = torch.where(drop.view(-1, 1), sd["null_prompt_cond"], pc)
    t = torch.cat((t, pc), dim=-1)
    c = perceiver_resampler(sub(sd, "perceiver_resampler."), masked, prompt_mask, cfg.heads)
    c = torch.where(drop.view(-1, 1, 1), sd["null_prompt_tokens"], c)
    h = causal_conv1d(x, sd["init_conv.weight"], sd["init_conv.bias"])
    h = wavenet(sub(sd, "wavenet."), h, cfg.wavenet_stacks, cfg.wavenet_layers, t)
    h = h + positional_embedding(mask, cfg.dim)
    h = transformer(sub(sd, "transformer."), h, cfg.depth, cfg.heads, mask, t, context=c)
    return F.linear(h, sd["final_proj.weight"], sd["final_proj.bias"])


def eps_forward_with_cond_scale(sd, cfg, x, times, mask, prompt, prompt_mask, cond_scale: float) -> Tensor:
    """forward_with_cond_scale latent_module.py:813-826."""
    B = x.shape[0]
    cond = eps_forward_cond(sd, cfg, x, times, mask, prompt, prompt_mask, torch.zeros(B, dtype=torch.bool))
    if cond_scale == 1.0:
        return cond
    null = eps_forward_cond(sd, cfg, x, times, mask, prompt, prompt_mask, torch.ones(B, dtype=torch.bool))
    return null + (cond - null) * cond_scale


# --------------------------------------------------------------------------- VAE
def vae_encode_params(sd: SD, cfg: VaeConfig, feat: Tensor) -> Tensor:
    """Encoder WaveNets of SpeechVAEEncoderDecoder.encode_feature latent_module.py:1099-1106.

    feat [B,T,768] -> posterior parameters [B,T,2*latent] ([mean ; logvar] on channels).
    """
    x = feat
    for n in range(len(cfg.chan_mults())):
        x = wavenet(sub(sd, f"encoder_wave.{n}."), x, cfg.stacks, cfg.layers)
    return x


def posterior_sample(params: Tensor, noise: Tensor):
    """DiagonalGaussianDistribution.__init__/sample distributions.py:24-41 (channels-last)."""
    mean, logvar = params.chunk(2, dim=-1)
    logvar = logvar.clamp(-30.0, 20.0)
    return mean + torch.exp(0.5 * logvar) * noise


def posterior_kl(params: Tensor, mask: Tensor) -> Tensor:
    """kl_3d distributions.py:62-74: pads zeroed but counted in the mean -> [B]."""
    mean, logvar = params.chunk(2, dim=-1)
    logvar = logvar.clamp(-30.0, 20.0)
    kl = mean.pow(2) + torch.exp(logvar) - 1.0 - logvar
    kl = kl.masked_fill(~mask.unsqueeze(-1), 0.0)
    return 0.5 * kl.mean(dim=[1, 2])


def vae_encode(sd: SD, cfg: VaeConfig, feat: Tensor, noise: Tensor) -> Tensor:
    """encode_feature latent_module.py:1099-1107 with injected posterior noise -> [B,T,latent]."""
    return posterior_sample(vae_encode_params(sd, cfg, feat), noise)


def vae_decode(sd: SD, cfg: VaeConfig, latent: Tensor, mask: Tensor):
    """decode_feature latent_module.py:1109-1116 -> (feature [B,T,768], logits [B,T,1004]).

    No positional embedding on this path.
    """
    x = latent
    for n in range(len(cfg.chan_mults())):
        x = wavenet(sub(sd, f"decoder_wave.{n}."), x, cfg.stacks, cfg.layers)
    dec = transformer(sub(sd, "decoder_tf."), x, cfg.depth, cfg.heads, mask, None, which="vae")
    return dec, F.linear(dec, sd["decoder_lm.weight"], sd["decoder_lm.bias"])


def vae_forward(sd: SD, cfg: VaeConfig, feat: Tensor, mask: Tensor, noise: Tensor):
    """SpeechVAEEncoderDecoder.forward latent_module.py:1118-1142 -> (mse, logits, kl)."""
    params = vae_encode_params(sd, cfg, feat)
    z = posterior_sample(params, noise)
    kl = posterior_kl(params, mask).mean()
    dec, logits = vae_decode(sd, cfg, z, mask)
    sel = mask.unsqueeze(2).expand(-1, -1, dec.shape[2])
    return F.mse_loss(dec[sel], feat[sel]), logits, kl


def vae_criterion(sd: SD, cfg: VaeConfig, feat: Tensor, units: Tensor, lengths: Tensor, noise: Tensor):
    """SpeechVAEDecoderLoss.forward fairseq/criterions/speech_vae_decoder_loss.py:45-95."""
    mask = lengths_to_mask(lengths, feat.shape[1])
    mse, logits, kl = vae_forward(sd, cfg, feat, mask, noise)
    lprobs = F.log_softmax(logits, dim=-1).view(-1, logits.shape[-1])
    tgt = units.reshape(-1)
    keep = tgt.ne(0)
    acc = (lprobs.argmax(1)[keep] == tgt[keep]).sum() / keep.sum()
    ntokens = int(lengths.sum())
    loss, nll = label_smoothed_nll_loss(lprobs, tgt, 0.1, 0)
    loss, nll = loss / ntokens, nll / ntokens
    total = 0.1 * loss + 10 * mse + 0.0001 * kl
    return {"loss": total, "nll_loss": nll, "mse_loss": mse, "kl_loss": kl, "acc": acc}


# --------------------------------------------------------------------------- scheduler
def cosine_betas(n: int, max_beta: float = 0.999) -> np.ndarray:
    """betas_for_alpha_bar + "cosine": latent_module.py:1145-1162, 1217-1221 (float64)."""
    ab = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    return np.array([min(1 - ab((i + 1) / n) / ab(i / n), max_beta) for i in range(n)])


def linear_betas(n: int) -> np.ndarray:
    """get_named_beta_schedule("linear") diffusion/gaussian_diffusion.py:98-118."""
    s = 1000 / n
    return np.linspace(s * 0.0001, s * 0.02, n, dtype=np.float64)


class ScheduleTables:
    """float64 tables of DDPMScheduler latent_module.py:1241-1276 and
    GaussianDiffusion.__init__ diffusion/gaussian_diffusion.py:144-201 (identical algebra)."""

    def __init__(self, betas: np.ndarray):
        betas = np.asarray(betas, dtype=np.float64)
        self.betas = betas
        self.num_timesteps = len(betas)
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        self.alphas_cumprod = ac
        self.alphas_cumprod_prev = np.append(1.0, ac[:-1])
        self.alphas_cumprod_next = np.append(ac[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(ac)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - ac)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - ac)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / ac)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / ac - 1)
        pv = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - ac)
        self.posterior_variance = pv
        self.posterior_log_variance_clipped = np.log(np.append(pv[1], pv[1:])) if len(pv) > 1 else np.array([])
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - ac)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - ac)

    def at(self, name: str, t: Tensor, ndim: int) -> Tensor:
        """_extract_into_tensor latent_module.py:1226-1238: gather, cast to fp32, broadcast."""
        arr = name if isinstance(name, np.ndarray) else getattr(self, name)
        v = torch.from_numpy(arr)[t].float()
        return v.view(-1, *([1] * (ndim - 1)))


def ddpm_tables(timesteps: int) -> ScheduleTables:
    """DDPMScheduler(timesteps): cosine schedule, latent_module.py:1241-1246."""
    return ScheduleTables(cosine_betas(timesteps))


# --------------------------------------------------------------------------- DDIM sampling
def ddim_update(tab: ScheduleTables, x: Tensor, eps: Tensor, t: Tensor) -> Tensor:
    """One eta=0 DDIM update, latent_module.py:1419-1442 (safe_div :958-959).

    All coefficients are the fp32 casts of the float64 tables; 1-abar_prev is formed in fp32.
    """
    sa = tab.at("sqrt_alphas_cumprod", t, x.ndim)
    s1 = tab.at("sqrt_one_minus_alphas_cumprod", t, x.ndim)
    x1 = (x - s1 * eps) / sa.clamp(min=1e-10)
    pn = (x - sa * x1) / s1.clamp(min=1e-10)
    abp = tab.at("alphas_cumprod_prev", t, x.ndim)
    return x1 * torch.sqrt(abp) + torch.sqrt(1 - abp) * pn


def ddim_sample(eps_sd: SD, eps_cfg: EpsConfig, vae_sd: SD, vae_cfg: VaeConfig, timesteps: int,
                feat: Tensor, mask: Tensor, ref_units: Tensor, start_step: int,
                post_noise: Tensor, start_noise: Tensor, return_trace: bool = False):
    """LatentDiscreteModel.ddim_sample latent_module.py:1385-1471 with injected noise.

    post_noise [B,T,latent]: VAE posterior noise (CPU RNG upstream, distributions.py:38);
    start_noise [B,T,latent]: the randn at :1409.  Model evaluated for t=start-1..1 (t=0
    only when start_step==1); returns (units list, match, total, recon[, trace]).
    """
    tab = ddpm_tables(timesteps)
    z = vae_encode(vae_sd, vae_cfg, feat, post_noise)
    B = z.shape[0]
    ts = torch.full((B,), start_step, dtype=torch.long)
    x = tab.at("sqrt_alphas_cumprod", ts, 3) * z + tab.at("sqrt_one_minus_alphas_cumprod", ts, 3) * start_noise
    trace = [x.clone()]
    for time in range(start_step - 1, -1, -1):
        t = torch.full((B,), time, dtype=torch.long)
        eps = eps_forward(eps_sd, eps_cfg, x, t, mask)
        x = ddim_update(tab, x, eps, t)
        trace.append(x.clone())
        if time == 1:
            break
    recon, logits = vae_decode(vae_sd, vae_cfg, x, mask)
    pred = logits.argmax(dim=-1) - 4
    match = int((pred[mask] == ref_units[mask]).sum())
    total = int(mask.sum())
    lens = mask.sum(dim=1)
    units = [pred[i, : int(lens[i])] for i in range(B)]
    if return_trace:
        return units, match, total, recon, trace
    return units, match, total, recon


# --------------------------------------------------------------------------- training forward
def diffusion_train_forward(eps_sd: SD, eps_cfg: EpsConfig, vae_sd: SD, vae_cfg: VaeConfig, timesteps: int,
                            feat: Tensor, units: Tensor, mask: Tensor, times: Tensor, post_noise: Tensor,
                            jitter_noise: Tensor, true_noise: Tensor, multitask: bool = True):
    """LatentDiscreteModel.forward latent_module.py:1514-1613 with injected t and noises."""
    tab = ddpm_tables(timesteps)
    z = vae_encode(vae_sd, vae_cfg, feat, post_noise)
    beta0 = tab.at("betas", torch.zeros_like(times), 3)
    x1 = z + jitter_noise * beta0  # beta_0, not sqrt(beta_0): :1534-1536
    sa = tab.at("sqrt_alphas_cumprod", times, 3)
    s1 = tab.at("sqrt_one_minus_alphas_cumprod", times, 3)
    xt = sa * x1 + s1 * true_noise
    eps = eps_forward(eps_sd, eps_cfg, xt, times, mask)
    sa1 = tab.at("sqrt_alphas_cumprod", times, 1)
    s11 = tab.at("sqrt_one_minus_alphas_cumprod", times, 1)
    snr = sa1 ** 2 / s11 ** 2
    weight = snr.clamp(max=5.0) / snr
    mse = (eps - true_noise) ** 2
    mse = mse.masked_fill(~mask.unsqueeze(2), 0.0).flatten(1).mean(dim=1)  # pads in the divisor
    noise_mse = (mse * weight).mean()
    x1_hat = (xt - s1 * eps) / sa.clamp(min=1e-10)
    dec, logits = vae_decode(vae_sd, vae_cfg, x1_hat, mask)
    sel = mask.unsqueeze(2).expand(-1, -1, dec.shape[2])
    recon_mse = F.mse_loss(dec[sel], feat[sel])
    lprobs = F.log_softmax(logits, dim=-1).view(-1, logits.shape[-1])
    tgt = units.reshape(-1)
    keep = tgt.ne(0)
    acc = (lprobs.argmax(1)[keep] == tgt[keep]).sum() / keep.sum()
    smooth, _ = label_smoothed_nll_loss(lprobs, tgt, 0.1, 0)
    smooth = smooth / keep.sum()
    recon = 50 * recon_mse + smooth
    total = noise_mse + recon / timesteps if multitask else noise_mse
    return {"total_loss": total, "nll_loss": smooth, "recon_mse_loss": recon_mse,
            "noise_loss": noise_mse, "acc": acc}


# --------------------------------------------------------------------------- generic Gaussian scheduler
def ddpm_chain(eps_sd: SD, eps_cfg: EpsConfig, timesteps: int, x: Tensor, mask: Tensor, start_step: int, noises: Tensor,
               var_type: str = "fixed_small", clip_denoised: bool = False) -> Tensor:
    """Ancestral (DDPM) chain over the eps-predictor on the cosine schedule: GaussianDiffusion.p_sample
    (diffusion/gaussian_diffusion.py:376-417) for t = start_step-1 .. 0 from x at index start_step-1, noises[k] used at
    t = start_step-1-k (what dn_ddpm_loop runs on the device; pinned by tests/golden/ddpm_chain.npz, the real reference's steps)."""
    diff = GaussianDiffusionOracle(cosine_betas(timesteps), var_type)
    model = lambda xx, tt: eps_forward(eps_sd, eps_cfg, xx, tt, mask)  # noqa: E731
    for k, t in enumerate(range(start_step - 1, -1, -1)):
        x = diff.p_sample(model, x, torch.full((x.shape[0],), t, dtype=torch.long), noises[k], clip_denoised)["sample"]
    return x


def space_timesteps(num_timesteps: int, section_counts) -> set:
    """diffusion/respace.py:12-62."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per, extra = divmod(num_timesteps, len(section_counts))
    start, steps = 0, []
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            steps.append(start + round(cur))
            cur += stride
        start += size
    return set(steps)


class GaussianDiffusionOracle:
    """Subset of GaussianDiffusion / SpacedDiffusion used by the north-star path
    (diffusion/gaussian_diffusion.py:144-560, diffusion/respace.py:65-129), eps-prediction.

    ``var_type`` in {"fixed_large", "fixed_small", "learned_range"}.  ``model`` is any
    callable (x, t) -> output with channels on dim 1 like upstream.
    """

    def __init__(self, betas: np.ndarray, var_type: str = "fixed_large", use_timesteps: Optional[Sequence[int]] = None):
        betas = np.asarray(betas, dtype=np.float64)
        self.timestep_map = list(range(len(betas)))
        if use_timesteps is not None:  # SpacedDiffusion.__init__ respace.py:72-88
            base = ScheduleTables(betas)
            keep, last, new, self.timestep_map = set(use_timesteps), 1.0, [], []
            for i, ac in enumerate(base.alphas_cumprod):
                if i in keep:
                    new.append(1 - ac / last)
                    last = ac
                    self.timestep_map.append(i)
            betas = np.array(new)
        self.tab = ScheduleTables(betas)
        self.var_type = var_type
        self.num_timesteps = self.tab.num_timesteps

    def _call(self, model, x, t):
        return model(x, torch.tensor(self.timestep_map, dtype=t.dtype)[t])  # _WrappedModel respace.py:117-129

    def q_sample(self, x0, t, noise):
        """gaussian_diffusion.py:215-230."""
        a = self.tab.at
        return a("sqrt_alphas_cumprod", t, x0.ndim) * x0 + a("sqrt_one_minus_alphas_cumprod", t, x0.ndim) * noise

    def q_posterior(self, x0, xt, t):
        """gaussian_diffusion.py:232-252."""
        a = self.tab.at
        mean = a("posterior_mean_coef1", t, xt.ndim) * x0 + a("posterior_mean_coef2", t, xt.ndim) * xt
        return mean, a("posterior_variance", t, xt.ndim), a("posterior_log_variance_clipped", t, xt.ndim)

    def predict_xstart_from_eps(self, xt, t, eps):
        """gaussian_diffusion.py:334-339."""
        a = self.tab.at
        return a("sqrt_recip_alphas_cumprod", t, xt.ndim) * xt - a("sqrt_recipm1_alphas_cumprod", t, xt.ndim) * eps

    def p_mean_variance(self, model, x, t, clip_denoised=True):
        """gaussian_diffusion.py:254-332 (EPSILON mean type)."""
        out = self._call(model, x, t)
        tab, a = self.tab, self.tab.at
        if self.var_type == "learned_range":
            C = x.shape[1]
            out, v = torch.split(out, C, dim=1)
            min_log = a("posterior_log_variance_clipped", t, x.ndim)
            max_log = a(np.log(tab.betas), t, x.ndim)
            frac = (v + 1) / 2
            logvar = frac * max_log + (1 - frac) * min_log
            var = torch.exp(logvar)
        elif self.var_type == "fixed_large":
            arr = np.append(tab.posterior_variance[1], tab.betas[1:])
            var = a(arr, t, x.ndim) + torch.zeros_like(x)
            logvar = a(np.log(arr), t, x.ndim) + torch.zeros_like(x)
        else:
            var = a("posterior_variance", t, x.ndim) + torch.zeros_like(x)
            logvar = a("posterior_log_variance_clipped", t, x.ndim) + torch.zeros_like(x)
        x0 = self.predict_xstart_from_eps(x, t, out)
        if clip_denoised:
            x0 = x0.clamp(-1, 1)
        mean, _, _ = self.q_posterior(x0, x, t)
        return {"mean": mean, "variance": var, "log_variance": logvar, "pred_xstart": x0}

    def p_sample(self, model, x, t, noise, clip_denoised=True):
        """gaussian_diffusion.py:376-417: mean + 1[t!=0]*exp(.5 logvar)*noise."""
        out = self.p_mean_variance(model, x, t, clip_denoised)
        nz = (t != 0).float().view(-1, *([1] * (x.ndim - 1)))
        return {"sample": out["mean"] + nz * torch.exp(0.5 * out["log_variance"]) * noise,
                "pred_xstart": out["pred_xstart"]}

    def ddim_sample(self, model, x, t, noise, clip_denoised=True, eta=0.0):
        """gaussian_diffusion.py:513-560."""
        out = self.p_mean_variance(model, x, t, clip_denoised)
        a = self.tab.at
        eps = (a("sqrt_recip_alphas_cumprod", t, x.ndim) * x - out["pred_xstart"]) / a("sqrt_recipm1_alphas_cumprod", t, x.ndim)
        ab, abp = a("alphas_cumprod", t, x.ndim), a("alphas_cumprod_prev", t, x.ndim)
        sigma = eta * torch.sqrt((1 - abp) / (1 - ab)) * torch.sqrt(1 - ab / abp)
        mean = out["pred_xstart"] * torch.sqrt(abp) + torch.sqrt(1 - abp - sigma ** 2) * eps
        nz = (t != 0).float().view(-1, *([1] * (x.ndim - 1)))
        return {"sample": mean + nz * sigma * noise, "pred_xstart": out["pred_xstart"]}

    def p_sample_loop(self, model, x_T, noises, clip_denoised=True):
        """gaussian_diffusion.py:419-511 with injected per-step noise (noises[i] used at step index i)."""
        x = x_T
        for i in range(self.num_timesteps - 1, -1, -1):
            t = torch.full((x.shape[0],), i, dtype=torch.long)
            x = self.p_sample(model, x, t, noises[i], clip_denoised)["sample"]
        return x

    def ddim_reverse_sample(self, model, x, t, clip_denoised=True):
        """gaussian_diffusion.py:562-598 (eta = 0): x_{t+1} along the deterministic path."""
        out = self.p_mean_variance(model, x, t, clip_denoised)
        a = self.tab.at
        eps = (a("sqrt_recip_alphas_cumprod", t, x.ndim) * x - out["pred_xstart"]) / a("sqrt_recipm1_alphas_cumprod", t, x.ndim)
        abn = a(np.append(self.tab.alphas_cumprod[1:], 0.0), t, x.ndim)  # alphas_cumprod_next (:166)
        return {"sample": out["pred_xstart"] * torch.sqrt(abn) + torch.sqrt(1 - abn) * eps, "pred_xstart": out["pred_xstart"]}

    def vb_terms_bpd(self, model, x0, xt, t, clip_denoised=True):
        """gaussian_diffusion.py:682-713 with diffusion_utils.py:10-88: KL(q(x_{t-1}|x_t,x_0) || p(x_{t-1}|x_t)) in bits per
        dimension, the discretised-Gaussian decoder NLL at t = 0."""
        true_mean, _, true_logvar = self.q_posterior(x0, xt, t)
        out = self.p_mean_variance(model, xt, t, clip_denoised)
        lv1, lv2 = true_logvar + torch.zeros_like(xt), out["log_variance"]
        kl = 0.5 * (-1.0 + lv2 - lv1 + torch.exp(lv1 - lv2) + ((true_mean - out["mean"]) ** 2) * torch.exp(-lv2))
        kl = kl.flatten(1).mean(dim=1) / np.log(2.0)
        cdf = lambda v: 0.5 * (1.0 + torch.tanh(np.sqrt(2.0 / np.pi) * (v + 0.044715 * torch.pow(v, 3))))
        centered = x0 - out["mean"]
        inv_stdv = torch.exp(-0.5 * lv2)
        cdf_plus, cdf_min = cdf(inv_stdv * (centered + 1.0 / 255.0)), cdf(inv_stdv * (centered - 1.0 / 255.0))
        log_probs = torch.where(x0 < -0.999, torch.log(cdf_plus.clamp(min=1e-12)),
                                torch.where(x0 > 0.999, torch.log((1.0 - cdf_min).clamp(min=1e-12)),
                                            torch.log((cdf_plus - cdf_min).clamp(min=1e-12))))
        nll = (-log_probs).flatten(1).mean(dim=1) / np.log(2.0)
        return {"output": torch.where(t == 0, nll, kl), "pred_xstart": out["pred_xstart"]}

    def training_losses(self, model, x0, t, noise, loss_type="mse"):
        """gaussian_diffusion.py:715-786, EPSILON mean type.  loss_type in {"mse", "rescaled_mse", "kl", "rescaled_kl"}; with a
        learned variance the MSE types add the VB term computed on the (detached) mean prediction."""
        xt = self.q_sample(x0, t, noise)
        if loss_type in ("kl", "rescaled_kl"):
            loss = self.vb_terms_bpd(model, x0, xt, t, clip_denoised=False)["output"]
            return {"loss": loss * self.num_timesteps if loss_type == "rescaled_kl" else loss}
        out = self._call(model, xt, t)
        terms = {}
        if self.var_type == "learned_range":
            frozen = out
            out = torch.split(out, x0.shape[1], dim=1)[0]
            terms["vb"] = self.vb_terms_bpd(lambda *a: frozen, x0, xt, t, clip_denoised=False)["output"]
            if loss_type == "rescaled_mse":
                terms["vb"] = terms["vb"] * self.num_timesteps / 1000.0
        terms["mse"] = ((noise - out) ** 2).flatten(1).mean(dim=1)
        terms["loss"] = terms["mse"] + terms["vb"] if "vb" in terms else terms["mse"]
        return terms

    def training_mse(self, model, x0, t, noise):
        """training_losses, MSE / EPSILON / fixed variance branch, gaussian_diffusion.py:715-786."""
        xt = self.q_sample(x0, t, noise)
        out = self._call(model, xt, t)
        return ((noise - out) ** 2).flatten(1).mean(dim=1)


def create_diffusion_oracle(timestep_respacing="", noise_schedule="linear", sigma_small=False,
                            learn_sigma=True, diffusion_steps=1000) -> GaussianDiffusionOracle:
    """create_diffusion diffusion/__init__.py:10-46 (eps-prediction, MSE)."""
    betas = linear_betas(diffusion_steps) if noise_schedule == "linear" else cosine_betas(diffusion_steps)
    if not timestep_respacing:
        timestep_respacing = [diffusion_steps]
    var = "learned_range" if learn_sigma else ("fixed_small" if sigma_small else "fixed_large")
    return GaussianDiffusionOracle(betas, var, space_timesteps(diffusion_steps, timestep_respacing))


# --------------------------------------------------------------------------- deterministic weights
def _unit_hash_normal(name: str, shape, scale: float) -> Tensor:
    g = torch.Generator().manual_seed(int.from_bytes(name.encode(), "little") % (2 ** 31 - 1))
    return torch.randn(*shape, generator=g) * scale


# The recipe-sized state dicts take 4-12 s to draw and the GPU test tier asks for the same ones once per arithmetic mode: the last few
# are kept (callers get a fresh dict over the SAME tensors -- they are read-only by convention: tests upload or load_state_dict them).
_SD_KEEP = 3
_sd_recent: "OrderedDict[str, SD]" = None  # type: ignore[assignment]


def _recent_state_dict(maker):
    import functools
    from collections import OrderedDict

    @functools.wraps(maker)
    def cached(cfg, seed: str = maker.__defaults__[0]):
        global _sd_recent
        if _sd_recent is None:
            _sd_recent = OrderedDict()
        key = f"{maker.__name__}|{cfg!r}|{seed}"
        if key in _sd_recent:
            _sd_recent.move_to_end(key)
        else:
            _sd_recent[key] = maker(cfg, seed)
            while len(_sd_recent) > _SD_KEEP:
                _sd_recent.popitem(last=False)
        return dict(_sd_recent[key])

    return cached


@_recent_state_dict
def make_eps_state_dict(cfg: EpsConfig, seed: str = "eps") -> SD:
    """Portable deterministic weights keyed by parameter name, in the reference's
    state-dict layout of ``Model``.  Scales follow PyTorch's default
    fan-in init (uniform variance 1/(3 fan_in)) so activations stay O(1) as upstream."""
    D, Z, C = cfg.dim, cfg.latent_dim, cfg.dim * cfg.dim_cond_mult
    inner = int(D * 4 * 2 / 3)
    hd = cfg.heads * cfg.dim_head
    sd: SD = {}
    cond = cfg.dim_prompt > 0
    C2 = C * (2 if cond else 1)  # the conditional variant concatenates the pooled-prompt condition to the time condition (:784)

    def lin(name, out, inp, bias=True, k=None):
        shape = (out, inp) if k is None else (out, inp, k)
        fan = inp * (k or 1)
        sd[name + ".weight"] = _unit_hash_normal(seed + name + ".w", shape, (1.0 / (3 * fan)) ** 0.5)
        if bias:
            sd[name + ".bias"] = _unit_hash_normal(seed + name + ".b", (out,), (1.0 / (3 * fan)) ** 0.5)

    lin("init_conv", D, Z, k=1)
    sd["to_time_cond.0.weights"] = _unit_hash_normal(seed + "ttc0", (D // 2,), 1.0)
    lin("to_time_cond.1", C, D + 1)
    lin("wavenet.init_conv", D, D, k=3)
    for s in range(cfg.wavenet_stacks):
        for i in range(cfg.wavenet_layers):
            p = f"wavenet.stacks.{s}.blocks.{i}."
            lin(p + "to_time_cond", 2 * D, C2)
            lin(p + "conv", D, D, k=3)
            lin(p + "res_conv", D, D, k=1)
            if s == cfg.wavenet_stacks - 1:
                lin(p + "skip_conv", D, D, k=1)
    lin("wavenet.final_conv", D, D, k=1)
    for layer in range(cfg.depth):
        p = f"transformer.layers.{layer}."
        lin(p + "0.to_gamma_beta", 2 * D, C2)
        lin(p + "1.to_q", hd, D, bias=False)
        lin(p + "1.to_kv", 2 * hd, D, bias=False)
        lin(p + "1.to_out", D, hd, bias=False)
        if cond:
            lin(p + "2.to_gamma_beta", 2 * D, C2)
            lin(p + "3.to_q", hd, D, bias=False)
            lin(p + "3.to_kv", 2 * hd, D, bias=False)
            lin(p + "3.to_out", D, hd, bias=False)
        lin(p + "4.to_gamma_beta", 2 * D, C2)
        lin(p + "5.0", 2 * inner, D)
        lin(p + "5.2.1", inner, inner, k=3)
        lin(p + "5.3", D, inner)
    sd["transformer.to_pred.0.gamma"] = 1.0 + _unit_hash_normal(seed + "tpg", (D,), 0.05)
    lin("transformer.to_pred.1", D, D, bias=False)
    lin("final_proj", Z, D)
    if cond:
        m, P = cfg.num_latents_m, cfg.dim_prompt
        sd["null_prompt_cond"] = _unit_hash_normal(seed + "npc", (C,), 0.02)
        sd["null_prompt_tokens"] = _unit_hash_normal(seed + "npt", (m, D), 0.02)
        lin("to_prompt_cond.1", C, P)
        r = "perceiver_resampler."
        sd[r + "latents"] = _unit_hash_normal(seed + "lat", (m, D), 0.02)
        lin(r + "proj_context", D, P)
        for l in range(cfg.resampler_depth):
            lin(r + f"layers.{l}.0.to_q", hd, D, bias=False)
            lin(r + f"layers.{l}.0.to_kv", 2 * hd, D, bias=False)
            lin(r + f"layers.{l}.0.to_out", D, hd, bias=False)
            lin(r + f"layers.{l}.1.0", 2 * inner, D)
            lin(r + f"layers.{l}.1.2", D, inner)
        sd[r + "norm.gamma"] = 1.0 + _unit_hash_normal(seed + "rng", (D,), 0.05)
    return sd


@_recent_state_dict
def make_vae_state_dict(cfg: VaeConfig, seed: str = "vae") -> SD:
    """Deterministic weights in the state-dict layout of ``SpeechVAEEncoderDecoder``."""
    sd: SD = {}

    def lin(name, out, inp, bias=True, k=None):
        shape = (out, inp) if k is None else (out, inp, k)
        fan = inp * (k or 1)
        sd[name + ".weight"] = _unit_hash_normal(seed + name + ".w", shape, (1.0 / (3 * fan)) ** 0.5)
        if bias:
            sd[name + ".bias"] = _unit_hash_normal(seed + name + ".b", (out,), (1.0 / (3 * fan)) ** 0.5)

    def wave(prefix, cin, cout):
        lin(prefix + "init_conv", cout, cin, k=3)
        for s in range(cfg.stacks):
            for i in range(cfg.layers):
                p = f"{prefix}stacks.{s}.blocks.{i}."
                lin(p + "conv", cout, cout, k=3)
                lin(p + "res_conv", cout, cout, k=1)
                if s == cfg.stacks - 1:
                    lin(p + "skip_conv", cout, cout, k=1)
        lin(prefix + "final_conv", cout, cout, k=1)

    cur = cfg.dim
    mults = cfg.chan_mults()
    for n, m in enumerate(mults):
        wave(f"encoder_wave.{n}.", cur, cur // m)
        cur //= m
    first = True
    for n, m in enumerate(reversed(mults)):
        tgt = cur * m
        if first:
            cur //= 2
            first = False
        wave(f"decoder_wave.{n}.", cur, tgt)
        cur = tgt
    D = cfg.dim
    inner = int(D * 4 * 2 / 3)
    hd = cfg.heads * cfg.dim_head
    for layer in range(cfg.depth):
        p = f"decoder_tf.layers.{layer}."
        sd[p + "0.gamma"] = 1.0 + _unit_hash_normal(seed + p + "0g", (D,), 0.05)
        lin(p + "1.to_q", hd, D, bias=False)
        lin(p + "1.to_kv", 2 * hd, D, bias=False)
        lin(p + "1.to_out", D, hd, bias=False)
        sd[p + "4.gamma"] = 1.0 + _unit_hash_normal(seed + p + "4g", (D,), 0.05)
        lin(p + "5.0", 2 * inner, D)
        lin(p + "5.2.1", inner, inner, k=3)
        lin(p + "5.3", D, inner)
    sd["decoder_tf.to_pred.0.gamma"] = 1.0 + _unit_hash_normal(seed + "tpg", (D,), 0.05)
    lin("decoder_tf.to_pred.1", D, D, bias=False)
    lin("decoder_lm", cfg.vocab, D)
    return sd
