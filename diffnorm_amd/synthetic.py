"""Random-init weights and synthetic inputs of the reference's shapes for benchmarks and smoke runs
(there is no network for checkpoints).  State dicts use the reference's key layout and
PyTorch's default fan-in initialisation scale, like freshly constructed reference modules.
"""
from types import SimpleNamespace
from typing import Dict

import torch


def eps_config(dim=512, latent_dim=128, depth=12, heads=8, dim_head=64, wavenet_layers=8, wavenet_stacks=4, dim_cond_mult=4,
               dim_prompt=0, num_latents_m=64, resampler_depth=2):
    """Hyper-parameters of the eps-predictor `Model` (reference latent_module.py:709-728, diff_discrete.py:83-84); dim_prompt > 0 =
    the conditional variant (use_cond=True: dim_prompt 768, 64 resampled latents, :1325-1333)."""
    return SimpleNamespace(dim=dim, latent_dim=latent_dim, depth=depth, heads=heads, dim_head=dim_head,
                           wavenet_layers=wavenet_layers, wavenet_stacks=wavenet_stacks, dim_cond_mult=dim_cond_mult,
                           dim_prompt=dim_prompt, num_latents_m=num_latents_m, resampler_depth=resampler_depth)


class _Init:
    def __init__(self, seed: int):
        self.g = torch.Generator().manual_seed(seed)
        self.sd: Dict[str, torch.Tensor] = {}

    def lin(self, name, out, inp, bias=True, k=None):
        shape = (out, inp) if k is None else (out, inp, k)
        bound = (1.0 / (inp * (k or 1))) ** 0.5  # kaiming_uniform(a=sqrt(5)) bound of nn.Linear / nn.Conv1d
        self.sd[name + ".weight"] = (torch.rand(*shape, generator=self.g) * 2 - 1) * bound
        if bias:
            self.sd[name + ".bias"] = (torch.rand(out, generator=self.g) * 2 - 1) * bound


def random_eps_state_dict(cfg, seed: int = 0) -> Dict[str, torch.Tensor]:
    D, Z, C = cfg.dim, cfg.latent_dim, cfg.dim * cfg.dim_cond_mult
    inner, hd = int(D * 4 * 2 / 3), cfg.heads * cfg.dim_head
    cond = getattr(cfg, "dim_prompt", 0) > 0
    C2 = C * (2 if cond else 1)
    it = _Init(seed)
    it.lin("init_conv", D, Z, k=1)
    it.sd["to_time_cond.0.weights"] = torch.randn(D // 2, generator=it.g)
    it.lin("to_time_cond.1", C, D + 1)
    it.lin("wavenet.init_conv", D, D, k=3)
    for s in range(cfg.wavenet_stacks):
        for i in range(cfg.wavenet_layers):
            p = f"wavenet.stacks.{s}.blocks.{i}."
            it.lin(p + "to_time_cond", 2 * D, C2)
            it.lin(p + "conv", D, D, k=3)
            it.lin(p + "res_conv", D, D, k=1)
            if s == cfg.wavenet_stacks - 1:
                it.lin(p + "skip_conv", D, D, k=1)
    it.lin("wavenet.final_conv", D, D, k=1)
    for l in range(cfg.depth):
        p = f"transformer.layers.{l}."
        it.lin(p + "0.to_gamma_beta", 2 * D, C2)
        it.lin(p + "1.to_q", hd, D, bias=False)
        it.lin(p + "1.to_kv", 2 * hd, D, bias=False)
        it.lin(p + "1.to_out", D, hd, bias=False)
        if cond:
            it.lin(p + "2.to_gamma_beta", 2 * D, C2)
            it.lin(p + "3.to_q", hd, D, bias=False)
            it.lin(p + "3.to_kv", 2 * hd, D, bias=False)
            it.lin(p + "3.to_out", D, hd, bias=False)
        it.lin(p + "4.to_gamma_beta", 2 * D, C2)
        it.lin(p + "5.0", 2 * inner, D)
        it.lin(p + "5.2.1", inner, inner, k=3)
        it.lin(p + "5.3", D, inner)
    it.sd["transformer.to_pred.0.gamma"] = torch.ones(D)
    it.lin("transformer.to_pred.1", D, D, bias=False)
    it.lin("final_proj", Z, D)
    if cond:  # reference latent_module.py:752-773, 416-440 (std 0.02 normal for the null condition / tokens / latents)
        m, P = cfg.num_latents_m, cfg.dim_prompt
        it.sd["null_prompt_cond"] = torch.randn(C, generator=it.g) * 0.02
        it.sd["null_prompt_tokens"] = torch.randn(m, D, generator=it.g) * 0.02
        it.lin("to_prompt_cond.1", C, P)
        r = "perceiver_resampler."
        it.sd[r + "latents"] = torch.randn(m, D, generator=it.g) * 0.02
        it.lin(r + "proj_context", D, P)
        for l in range(cfg.resampler_depth):
            it.lin(r + f"layers.{l}.0.to_q", hd, D, bias=False)
            it.lin(r + f"layers.{l}.0.to_kv", 2 * hd, D, bias=False)
            it.lin(r + f"layers.{l}.0.to_out", D, hd, bias=False)
            it.lin(r + f"layers.{l}.1.0", 2 * inner, D)
            it.lin(r + f"layers.{l}.1.2", D, inner)
        it.sd[r + "norm.gamma"] = torch.ones(D)
    return it.sd


def random_vae_state_dict(dim=768, latent_dim=128, depth=6, heads=8, dim_head=96, stacks=2, layers=3, vocab=1004,
                          seed: int = 1) -> Dict[str, torch.Tensor]:
    from .packing import vae_mults

    it = _Init(seed)

    def wave(prefix, cin, cout):
        it.lin(prefix + "init_conv", cout, cin, k=3)
        for s in range(stacks):
            for i in range(layers):
                p = f"{prefix}stacks.{s}.blocks.{i}."
                it.lin(p + "conv", cout, cout, k=3)
                it.lin(p + "res_conv", cout, cout, k=1)
                if s == stacks - 1:
                    it.lin(p + "skip_conv", cout, cout, k=1)
        it.lin(prefix + "final_conv", cout, cout, k=1)

    mults = vae_mults(latent_dim)
    cur = dim
    for n, m in enumerate(mults):
        wave(f"encoder_wave.{n}.", cur, cur // m)
        cur //= m
    first = True
    for n, m in enumerate(reversed(mults)):
        tgt = cur * m
        wave(f"decoder_wave.{n}.", cur // 2 if first else cur, tgt)
        first = False
        cur = tgt
    inner, hd = int(dim * 4 * 2 / 3), heads * dim_head
    for l in range(depth):
        p = f"decoder_tf.layers.{l}."
        it.sd[p + "0.gamma"] = torch.ones(dim)
        it.lin(p + "1.to_q", hd, dim, bias=False)
        it.lin(p + "1.to_kv", 2 * hd, dim, bias=False)
        it.lin(p + "1.to_out", dim, hd, bias=False)
        it.sd[p + "4.gamma"] = torch.ones(dim)
        it.lin(p + "5.0", 2 * inner, dim)
        it.lin(p + "5.2.1", inner, inner, k=3)
        it.lin(p + "5.3", dim, inner)
    it.sd["decoder_tf.to_pred.0.gamma"] = torch.ones(dim)
    it.lin("decoder_tf.to_pred.1", dim, dim, bias=False)
    it.lin("decoder_lm", vocab, dim)
    return it.sd


def eps_step_flops(B: int, T: int) -> float:
    """Algorithmic FLOPs of one denoising step of the recipe-size eps-predictor (MAC = 2 FLOP)."""
    return B * T * (283_824_136 + 24_576 * T) + B * 236_982_272


def random_nar_decoder_state_dict(dim=512, ffn=2048, layers=6, vocab=1004, pad=1, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random-init weights of the NAR S2UT decoder under the reference's parameter names (`decoder.*` of NARS2UTTransformerModel's
    state dict, research/TranSpeech/nar_transformer.py:84-122 on fairseq's TransformerDecoder): nar_s2ut_transformer's sizes by default."""
    g = torch.Generator().manual_seed(seed)
    n = lambda *shape, scale=1.0: torch.randn(*shape, generator=g) * scale
    sd = {"embed_tokens.weight": n(vocab, dim, scale=dim ** -0.5), "embed_length.weight": n(256, dim, scale=dim ** -0.5)}
    sd["embed_tokens.weight"][pad] = 0
    for l in range(layers):
        p = f"layers.{l}."
        for att in ("self_attn", "encoder_attn"):
            for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):
                sd[p + f"{att}.{proj}.weight"], sd[p + f"{att}.{proj}.bias"] = n(dim, dim, scale=dim ** -0.5), n(dim, scale=0.02)
            sd[p + att + "_layer_norm.weight"], sd[p + att + "_layer_norm.bias"] = torch.ones(dim), torch.zeros(dim)
        sd[p + "fc1.weight"], sd[p + "fc1.bias"] = n(ffn, dim, scale=dim ** -0.5), n(ffn, scale=0.02)
        sd[p + "fc2.weight"], sd[p + "fc2.bias"] = n(dim, ffn, scale=ffn ** -0.5), n(dim, scale=0.02)
        sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"] = torch.ones(dim), torch.zeros(dim)
    sd["layer_norm.weight"], sd["layer_norm.bias"] = torch.ones(dim), torch.zeros(dim)
    sd["output_projection.weight"] = n(vocab, dim, scale=dim ** -0.5)
    return sd


def random_nar_encoder_state_dict(input_dim=80, conv_channels=1024, kernel=5, dim=512, ffn=2048, layers=12, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random-init weights of the NAR S2UT model's speech encoder under the reference's parameter names (S2TTransformerEncoder.state_dict():
    fairseq/models/speech_to_text/s2t_transformer.py:299-343): nar_s2ut_transformer's sizes by default (research/TranSpeech/nar_transformer.py:954-970)."""
    g = torch.Generator().manual_seed(seed)
    n = lambda *shape, scale=1.0: torch.randn(*shape, generator=g) * scale
    sd = {"subsample.conv_layers.0.weight": n(conv_channels, input_dim, kernel, scale=(input_dim * kernel) ** -0.5),
          "subsample.conv_layers.0.bias": n(conv_channels, scale=0.02),
          "subsample.conv_layers.1.weight": n(2 * dim, conv_channels // 2, kernel, scale=(conv_channels // 2 * kernel) ** -0.5),
          "subsample.conv_layers.1.bias": n(2 * dim, scale=0.02)}
    for l in range(layers):
        p = f"transformer_layers.{l}."
        for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):
            sd[p + f"self_attn.{proj}.weight"], sd[p + f"self_attn.{proj}.bias"] = n(dim, dim, scale=dim ** -0.5), n(dim, scale=0.02)
        sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"] = torch.ones(dim), torch.zeros(dim)
        sd[p + "fc1.weight"], sd[p + "fc1.bias"] = n(ffn, dim, scale=dim ** -0.5), n(ffn, scale=0.02)
        sd[p + "fc2.weight"], sd[p + "fc2.bias"] = n(dim, ffn, scale=ffn ** -0.5), n(dim, scale=0.02)
        sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"] = torch.ones(dim), torch.zeros(dim)
    sd["layer_norm.weight"], sd["layer_norm.bias"] = torch.ones(dim), torch.zeros(dim)
    return sd
