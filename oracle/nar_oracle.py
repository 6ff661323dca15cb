"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (diffnorm_amd/).

CPU restatement (plain torch fp32) of the model that runs inside SURVEY 8 f4's mask-predict loop: the decoder side of the
reference's NAR S2UT model, `NARS2UTTransformerModel` (research/TranSpeech/nar_transformer.py:569-976) --
  * TransformerUnitDecoder.forward (:321-420) = fairseq's TransformerDecoder.extract_features_scriptable
    (fairseq/models/transformer/transformer_decoder.py:219-330, full_context_alignment=True: no causal mask): scaled token
    embedding + sinusoidal positions (padding_idx-based, fairseq/utils.py:256-266), pre-norm layers of self-attention, encoder
    attention and ReLU FFN (fairseq/modules/transformer_layer.py:389-520; attention arithmetic fairseq/modules/multihead_attention.py:
    q scaled by d_h^-0.5, key padding mask, biases everywhere), final LayerNorm, the output projection, log_softmax;
  * forward_length / forward_length_prediction (:436-480): masked mean pooling of the encoder output, a linear map onto 256
    lengths, arg-max;
  * forward_decoder (:791-842), initialize_output_tokens (:844-885), regenerate_length_beam (:887-912).
Pinned by tests/golden/nar_decoder.npz: the REAL classes' outputs (oracle/gen_golden_nar.py)."""
import math
from collections import namedtuple
from dataclasses import dataclass
from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
DecoderOut = namedtuple("IterativeRefinementDecoderOut", ["output_tokens", "output_scores", "attn", "step", "max_step", "history"])


@dataclass
class NarConfig:
    """nar_s2ut_transformer (:954-1002): embed 512, FFN 2048, 6 layers, 8 heads, vocabulary = unit dictionary (1000 + 4 specials)."""
    embed_dim: int = 512
    ffn_dim: int = 2048
    layers: int = 6
    heads: int = 8
    vocab: int = 1004
    max_positions: int = 1024
    pad: int = 1
    unk: int = 3
    bos: int = 0
    eos: int = 2


def _unit_hash_normal(name: str, shape, scale: float) -> torch.Tensor:
    g = torch.Generator().manual_seed(int.from_bytes(name.encode(), "little") % (2 ** 31 - 1))
    return torch.randn(*shape, generator=g) * scale


def make_nar_state_dict(cfg: NarConfig, seed: str = "nar") -> SD:
    """Portable deterministic weights under the reference decoder's parameter names (`decoder.*` of the model's state dict)."""
    D, Fd, V = cfg.embed_dim, cfg.ffn_dim, cfg.vocab
    sd: SD = {}

    def lin(name, out, inp, bias=True, gain=1.0):
        sd[name + ".weight"] = _unit_hash_normal(seed + name + ".w", (out, inp), gain * inp ** -0.5)
        if bias:
            sd[name + ".bias"] = _unit_hash_normal(seed + name + ".b", (out,), 0.1)

    emb = _unit_hash_normal(seed + "emb", (V, D), D ** -0.5)
    emb[cfg.pad] = 0
    sd["embed_tokens.weight"] = emb
    sd["embed_length.weight"] = _unit_hash_normal(seed + "len", (256, D), D ** -0.5)
    for l in range(cfg.layers):
        p = f"layers.{l}."
        for att in ("self_attn", "encoder_attn"):
            for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):
                lin(p + att + "." + proj, D, D)
            sd[p + att + "_layer_norm.weight"] = 1.0 + _unit_hash_normal(seed + p + att + "lnw", (D,), 0.1)
            sd[p + att + "_layer_norm.bias"] = _unit_hash_normal(seed + p + att + "lnb", (D,), 0.1)
        lin(p + "fc1", Fd, D)
        lin(p + "fc2", D, Fd)
        sd[p + "final_layer_norm.weight"] = 1.0 + _unit_hash_normal(seed + p + "flnw", (D,), 0.1)
        sd[p + "final_layer_norm.bias"] = _unit_hash_normal(seed + p + "flnb", (D,), 0.1)
    sd["layer_norm.weight"] = 1.0 + _unit_hash_normal(seed + "lnw", (D,), 0.1)
    sd["layer_norm.bias"] = _unit_hash_normal(seed + "lnb", (D,), 0.1)
    lin("output_projection", V, D, bias=False, gain=2.0)
    return sd


def sinusoidal_table(num: int, dim: int, padding_idx: int) -> torch.Tensor:
    """SinusoidalPositionalEmbedding.get_embedding (fairseq/modules/sinusoidal_positional_embedding.py:36-58)."""
    half = dim // 2
    f = torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1)))
    ang = torch.arange(num, dtype=torch.float).unsqueeze(1) * f.unsqueeze(0)
    tab = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1).view(num, -1)
    if dim % 2 == 1:
        tab = torch.cat([tab, torch.zeros(num, 1)], dim=1)
    tab[padding_idx, :] = 0
    return tab


def _attention(sd: SD, prefix: str, q_in, kv_in, key_pad, heads: int):
    """MultiheadAttention.forward (the torch functional path): q scaled by d_h^-0.5, keys masked with -inf where key_pad."""
    B, Tq, D = q_in.shape
    dh = D // heads
    q = F.linear(q_in, sd[prefix + "q_proj.weight"], sd[prefix + "q_proj.bias"]) * dh ** -0.5
    k = F.linear(kv_in, sd[prefix + "k_proj.weight"], sd[prefix + "k_proj.bias"])
    v = F.linear(kv_in, sd[prefix + "v_proj.weight"], sd[prefix + "v_proj.bias"])
    split = lambda t: t.view(B, -1, heads, dh).transpose(1, 2)
    w = split(q) @ split(k).transpose(-1, -2)
    if key_pad is not None:
        w = w.masked_fill(key_pad[:, None, None, :], float("-inf"))
    o = (torch.softmax(w, dim=-1) @ split(v)).transpose(1, 2).reshape(B, Tq, D)
    return F.linear(o, sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"])


def decoder_logits(sd: SD, cfg: NarConfig, tokens: torch.Tensor, enc_out: torch.Tensor, enc_pad: torch.Tensor, normalize: bool = True):
    """tokens int64 [B,T]; enc_out fp32 [S,B,D] (the encoder's layout); enc_pad bool [B,S] (True = padding) or None
    -> log-probabilities (normalize) or logits [B,T,V]."""
    D = cfg.embed_dim
    nonpad = tokens.ne(cfg.pad)
    positions = (torch.cumsum(nonpad.long(), dim=1) * nonpad.long()) + cfg.pad  # fairseq/utils.py:256-266
    tab = sinusoidal_table(cfg.pad + 1 + tokens.size(1), D, cfg.pad)
    x = math.sqrt(D) * sd["embed_tokens.weight"][tokens] + tab[positions]
    self_pad = tokens.eq(cfg.pad)
    self_pad = self_pad if bool(self_pad.any()) else None  # (:283-284)
    mem = enc_out.transpose(0, 1)
    ln = lambda t, p: F.layer_norm(t, (D,), sd[p + ".weight"], sd[p + ".bias"], 1e-5)
    for l in range(cfg.layers):
        p = f"layers.{l}."
        x = x + _attention(sd, p + "self_attn.", ln(x, p + "self_attn_layer_norm"), ln(x, p + "self_attn_layer_norm"), self_pad, cfg.heads)
        x = x + _attention(sd, p + "encoder_attn.", ln(x, p + "encoder_attn_layer_norm"), mem, enc_pad, cfg.heads)
        h = ln(x, p + "final_layer_norm")
        x = x + F.linear(F.relu(F.linear(h, sd[p + "fc1.weight"], sd[p + "fc1.bias"])), sd[p + "fc2.weight"], sd[p + "fc2.bias"])
    x = ln(x, "layer_norm")
    out = F.linear(x, sd["output_projection.weight"])
    return F.log_softmax(out, -1) if normalize else out


# --------------------------------------------------------------------------- the speech encoder (round 4)
@dataclass
class NarEncoderConfig:
    """S2TTransformerEncoder of nar_s2ut_transformer (research/TranSpeech/nar_transformer.py:954-970): 80 fbank features, Conv1dSubsampler
    (kernels 5,5; 1024 mid channels), 512 / 2048 / 12 pre-norm layers / 8 heads."""
    input_dim: int = 80
    conv_channels: int = 1024
    kernel_sizes: tuple = (5, 5)
    embed_dim: int = 512
    ffn_dim: int = 2048
    layers: int = 12
    heads: int = 8
    pad: int = 1  # S2TTransformerEncoder.padding_idx (s2t_transformer.py:311)


def make_nar_encoder_state_dict(cfg: NarEncoderConfig, seed: str = "narenc") -> SD:
    """Portable deterministic weights under the reference encoder's parameter names (S2TTransformerEncoder.state_dict())."""
    D, Fd = cfg.embed_dim, cfg.ffn_dim
    sd: SD = {}

    def lin(name, out, inp):
        sd[name + ".weight"] = _unit_hash_normal(seed + name + ".w", (out, inp), inp ** -0.5)
        sd[name + ".bias"] = _unit_hash_normal(seed + name + ".b", (out,), 0.1)

    n = len(cfg.kernel_sizes)
    for i, k in enumerate(cfg.kernel_sizes):  # Conv1dSubsampler (fairseq/models/speech_to_text/modules/convolution.py:31-43)
        cin = cfg.input_dim if i == 0 else cfg.conv_channels // 2
        cout = cfg.conv_channels if i < n - 1 else D * 2
        sd[f"subsample.conv_layers.{i}.weight"] = _unit_hash_normal(seed + f"conv{i}.w", (cout, cin, k), 1.7 * (cin * k) ** -0.5)
        sd[f"subsample.conv_layers.{i}.bias"] = _unit_hash_normal(seed + f"conv{i}.b", (cout,), 0.1)
    for l in range(cfg.layers):
        p = f"transformer_layers.{l}."
        for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):
            lin(p + "self_attn." + proj, D, D)
        sd[p + "self_attn_layer_norm.weight"] = 1.0 + _unit_hash_normal(seed + p + "alnw", (D,), 0.1)
        sd[p + "self_attn_layer_norm.bias"] = _unit_hash_normal(seed + p + "alnb", (D,), 0.1)
        lin(p + "fc1", Fd, D)
        lin(p + "fc2", D, Fd)
        sd[p + "final_layer_norm.weight"] = 1.0 + _unit_hash_normal(seed + p + "flnw", (D,), 0.1)
        sd[p + "final_layer_norm.bias"] = _unit_hash_normal(seed + p + "flnb", (D,), 0.1)
    sd["layer_norm.weight"] = 1.0 + _unit_hash_normal(seed + "lnw", (D,), 0.1)
    sd["layer_norm.bias"] = _unit_hash_normal(seed + "lnb", (D,), 0.1)
    return sd


def subsampled_lengths(lengths: torch.Tensor, n_layers: int) -> torch.Tensor:
    """Conv1dSubsampler.get_out_seq_lens_tensor (convolution.py:45-49): floor((len - 1) / 2 + 1) per stride-2 layer."""
    out = lengths.clone()
    for _ in range(n_layers):
        out = ((out.float() - 1) / 2 + 1).floor().long()
    return out


def encoder_forward(sd: SD, cfg: NarEncoderConfig, feats: torch.Tensor, src_lengths: torch.Tensor):
    """S2TTransformerEncoder._forward (fairseq/models/speech_to_text/s2t_transformer.py:345-373), which
    S2STransformerEncoder.forward (research/TranSpeech/nar_transformer.py:56-76) calls with no speaker embedding:
    feats fp32 [B, L, input_dim], src_lengths [B] -> (encoder_out [S, B, D], encoder_padding_mask bool [B, S], lengths [B]).
    The subsampler convolves the PADDED batch (frames behind an utterance's end take part, as upstream); the transformer masks keys."""
    D = cfg.embed_dim
    x = feats.transpose(1, 2)  # B x C x L (convolution.py:53)
    for i, k in enumerate(cfg.kernel_sizes):
        x = F.conv1d(x, sd[f"subsample.conv_layers.{i}.weight"], sd[f"subsample.conv_layers.{i}.bias"], stride=2, padding=k // 2)
        x = F.glu(x, dim=1)
    x = x.transpose(1, 2)  # B x S x D
    lens = subsampled_lengths(src_lengths, len(cfg.kernel_sizes))
    S = x.size(1)
    pad = torch.arange(S)[None, :] >= lens[:, None]  # lengths_to_padding_mask (data_utils.py); S == max(lens) when one utterance fills the batch
    nonpad = (~pad).long()
    positions = torch.cumsum(nonpad, dim=1) * nonpad + cfg.pad  # make_positions on the mask itself (:350; fairseq/utils.py:256-266)
    x = math.sqrt(D) * x + sinusoidal_table(cfg.pad + 1 + S, D, cfg.pad)[positions]
    key_pad = pad if bool(pad.any()) else None
    ln = lambda t, p: F.layer_norm(t, (D,), sd[p + ".weight"], sd[p + ".bias"], 1e-5)
    for l in range(cfg.layers):  # TransformerEncoderLayerBase.forward, normalize_before (fairseq/modules/transformer_layer.py:163-226)
        p = f"transformer_layers.{l}."
        h = ln(x, p + "self_attn_layer_norm")
        x = x + _attention(sd, p + "self_attn.", h, h, key_pad, cfg.heads)
        h = ln(x, p + "final_layer_norm")
        x = x + F.linear(F.relu(F.linear(h, sd[p + "fc1.weight"], sd[p + "fc1.bias"])), sd[p + "fc2.weight"], sd[p + "fc2.bias"])
    x = ln(x, "layer_norm")
    return x.transpose(0, 1), pad, lens


def predict_lengths(sd: SD, enc_out: torch.Tensor, enc_pad) -> torch.Tensor:
    """forward_length + forward_length_prediction (:436-480) without an offset: masked mean of the encoder output -> 256-way arg-max."""
    if enc_pad is not None:
        keep = (~enc_pad).transpose(0, 1).type_as(enc_out)
        feats = ((enc_out / keep.sum(0)[None, :, None]) * keep[:, :, None]).sum(0)  # _mean_pooling, nonautoregressive_transformer.py:20-34
    else:
        feats = enc_out.mean(0)
    return F.log_softmax(F.linear(feats, sd["embed_length.weight"]), -1).max(-1)[1]


def blank_tokens(lengths: torch.Tensor, cfg: NarConfig) -> torch.Tensor:
    """(:861-868) unk for idx < length, pad after; lengths clamped to >= 2."""
    lengths = lengths.clamp(min=2)
    idx = torch.arange(int(lengths.max()), device=lengths.device)
    return torch.full((lengths.size(0), idx.numel()), cfg.pad, dtype=torch.long, device=lengths.device).masked_fill(
        idx[None, :] < lengths[:, None], cfg.unk)


def skeptical_unmasking(scores, nonpad, p):
    """fairseq/models/nat/cmlm_transformer.py:19-25."""
    order = scores.sort(-1)[1]
    boundary = ((nonpad.sum(1, keepdim=True).type_as(scores) - 2) * p).long()
    cut = torch.arange(scores.size(1), device=scores.device)[None, :] < boundary
    return torch.zeros_like(cut).scatter(1, order, cut)


class _Encoder:
    """Stand-in for the speech encoder (out of this row's scope: its output is a given tensor): returns what it was given and
    re-orders it like S2TTransformerEncoder.reorder_encoder_out (fairseq/models/speech_to_text/s2t_transformer.py:383-411)."""

    def __call__(self, enc, src_lengths=None):
        return enc

    def reorder_encoder_out(self, enc, order):
        order = order.reshape(-1)
        return {"encoder_out": [x.index_select(1, order) for x in enc["encoder_out"]],
                "encoder_padding_mask": [x.index_select(0, order) for x in enc["encoder_padding_mask"]],
                "encoder_embedding": [], "encoder_states": [], "src_tokens": [], "src_lengths": []}


class OracleNarModel:
    """The NAT interface of NARS2UTTransformerModel over `decoder_logits` (what the research generator drives)."""
    allow_length_beam = True

    def __init__(self, sd: SD, cfg: NarConfig):
        self.sd, self.cfg, self.encoder = sd, cfg, _Encoder()
        self.unk, self.pad = cfg.unk, cfg.pad

    def eval(self):
        return self

    def forward_encoder(self, encoder_inputs):
        return self.encoder(*encoder_inputs)

    @staticmethod
    def _pad_mask(enc):
        return enc["encoder_padding_mask"][0] if len(enc["encoder_padding_mask"]) > 0 else None

    def initialize_output_tokens(self, encoder_out, src_lengths, true_length=None):
        lengths = true_length if true_length is not None else predict_lengths(self.sd, encoder_out["encoder_out"][0], self._pad_mask(encoder_out)).long()
        tok = blank_tokens(lengths, self.cfg)
        return DecoderOut(tok, torch.zeros(tok.shape, dtype=encoder_out["encoder_out"][0].dtype, device=tok.device), None, 0, 0, None)

    def regenerate_length_beam(self, decoder_out, beam_size):
        lengths = decoder_out.output_tokens.ne(self.pad).sum(1)
        lengths = (lengths[:, None] + torch.arange(beam_size, device=lengths.device)[None, :] - beam_size // 2).view(-1)
        tok = blank_tokens(lengths, self.cfg)
        return decoder_out._replace(output_tokens=tok, output_scores=torch.zeros(tok.shape, dtype=decoder_out.output_scores.dtype, device=tok.device))

    def forward_decoder(self, decoder_out, encoder_out, decoding_format=None, **kwargs):
        step, max_step = decoder_out.step, decoder_out.max_step
        tokens, scores, history = decoder_out.output_tokens.clone(), decoder_out.output_scores.clone(), decoder_out.history
        masks = tokens.eq(self.unk)
        sc, tk = decoder_logits(self.sd, self.cfg, tokens, encoder_out["encoder_out"][0], self._pad_mask(encoder_out)).max(-1)
        tokens = torch.where(masks, tk, tokens)
        scores = torch.where(masks, sc, scores)
        if history is not None:
            history.append(tokens.clone())
        if (step + 1) < max_step:
            sk = skeptical_unmasking(scores, tokens.ne(self.pad), 1 - (step + 1) / max_step)
            tokens = tokens.masked_fill(sk, self.unk)
            scores = scores.masked_fill(sk, 0.0)
            if history is not None:
                history.append(tokens.clone())
        return decoder_out._replace(output_tokens=tokens, output_scores=scores, attn=None, history=history)
