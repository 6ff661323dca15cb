#!/usr/bin/env python3
"""Golden vectors for the speech ENCODER of SURVEY 8 f4's model (round 4): the REAL S2STransformerEncoder
(research/TranSpeech/nar_transformer.py:40-76) over fairseq's S2TTransformerEncoder (fairseq/models/speech_to_text/s2t_transformer.py:
295-385: Conv1dSubsampler -> sqrt(D) scaling -> sinusoidal positions of the padding mask -> pre-norm TransformerEncoderLayers ->
LayerNorm), leaf-loaded by oracle/ref_loader.load_reference_nar, on a small configuration with the portable weights of
nar_oracle.make_nar_encoder_state_dict (load_state_dict(strict=True) up to the two position / version buffers: pins the names):
a ragged batch of fbank-like features -> encoder_out [S, B, D], the padding mask and the subsampled lengths; and the whole
generate() of the research IterativeRefinementGenerator from [B, L, 80] features through encoder AND decoder.
Run in the build container only: python oracle/gen_golden_nar_encoder.py"""
import os
import sys
import types

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import nar_oracle as N  # noqa: E402
import ref_loader  # noqa: E402
from gen_golden import save  # noqa: E402
from gen_golden_configs import seeded  # noqa: E402
from gen_golden_nar import build_reference_model  # noqa: E402
from gen_golden_nar_configs import ENC_CFG  # noqa: E402


def build_reference_encoder(R):
    c = ENC_CFG
    args = types.SimpleNamespace(encoder_freezing_updates=0, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, activation_fn="relu",
                                 encoder_embed_dim=c.embed_dim, encoder_ffn_embed_dim=c.ffn_dim, encoder_layers=c.layers,
                                 encoder_attention_heads=c.heads, encoder_normalize_before=True, no_scale_embedding=False,
                                 conv_version="s2t_transformer", input_feat_per_channel=c.input_dim, input_channels=1, conv_channels=c.conv_channels,
                                 conv_kernel_sizes=",".join(str(k) for k in c.kernel_sizes), max_source_positions=6000, target_speaker_embed=False,
                                 decoder_embed_dim=c.embed_dim, decoder_ffn_embed_dim=c.ffn_dim, decoder_layers=1, decoder_attention_heads=c.heads)
    enc = R.nar.S2STransformerEncoder(args)
    sd = N.make_nar_encoder_state_dict(c, "narenc")
    missing, unexpected = enc.load_state_dict(sd, strict=False)
    assert set(missing) <= {"embed_positions._float_tensor", "version"} and not unexpected, (missing, unexpected)
    assert {n for n, _ in enc.named_parameters()} == set(sd), "parameter names differ from nar_oracle.make_nar_encoder_state_dict"
    enc.eval()
    return enc


def main():
    R = ref_loader.load_reference_nar()
    enc = build_reference_encoder(R)
    out = {}
    B, L = 4, 61
    lens = torch.tensor([61, 40, 23, 9])
    feats = seeded((B, L, ENC_CFG.input_dim), 901)
    feats = feats * (torch.arange(L)[None, :, None] < lens[:, None, None])  # the collater zero-pads
    with torch.no_grad():
        eo = enc(feats, lens)
    out.update(lens=lens, encoder_out=eo["encoder_out"][0], padding_mask=eo["encoder_padding_mask"][0],
               out_lens=enc.subsample.get_out_seq_lens_tensor(lens))
    # generate() end to end: features -> encoder -> length prediction -> mask-predict refinement with the decoder of nar_decoder.npz
    model, d = build_reference_model(R)
    model.encoder = enc
    model.forward_encoder = lambda inputs: enc(*inputs)  # NARS2UTTransformerModel.forward_encoder (:566-567)
    orig_to = torch.Tensor.to

    def to_cpu(self, *a, **k):
        if k.get("device") == "cuda":
            k = dict(k, device="cpu")
        return orig_to(self, *a, **k)

    torch.Tensor.to = to_cpu
    try:
        gen = R.gen.IterativeRefinementGenerator(d, max_iter=4, beam_size=1, adaptive=True)
        with torch.no_grad():
            hyps = gen.generate([model], {"net_input": {"src_tokens": feats, "src_lengths": lens}})
        out["gen_n"] = len(hyps)
        for i, h in enumerate(hyps):
            out[f"gen_h{i}_tokens"], out[f"gen_h{i}_scores"], out[f"gen_h{i}_steps"] = h[0]["tokens"], h[0]["positional_scores"], h[0]["steps"]
    finally:
        torch.Tensor.to = orig_to
    save("nar_encoder", **out)


if __name__ == "__main__":
    main()
