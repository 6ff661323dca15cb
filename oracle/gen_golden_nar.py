#!/usr/bin/env python3
"""Golden vectors for SURVEY 8 f4's model-in-the-loop: the REAL reference classes (leaf-loaded by oracle/ref_loader.load_reference_nar)
-- TransformerUnitDecoder (research/TranSpeech/nar_transformer.py:84-480) built on fairseq's TransformerDecoder /
TransformerDecoderLayer / MultiheadAttention, NARS2UTTransformerModel.{forward_decoder, initialize_output_tokens,
regenerate_length_beam} (:791-912) and the research IterativeRefinementGenerator (research/TranSpeech/
iterative_refinement_generator.py) -- on a small decoder with the portable weights of nar_oracle.make_nar_state_dict (loaded with
load_state_dict(strict=True): pins the parameter names), a given encoder output and ragged source lengths:
  logits / log-probabilities of one decoder pass on a partially masked ragged batch, the predicted lengths, and every hypothesis
  (tokens, scores, steps, history) of the generator for four settings incl. a length beam of 3.
The speech ENCODER is out of scope (its output is a given tensor): a stand-in with the reference's reorder_encoder_out semantics.
Run in the build container only: python oracle/gen_golden_nar.py"""
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

from gen_golden_nar_configs import CFG, SETTINGS, Dict1004, encoder_out  # noqa: E402


def build_reference_model(R):
    args = types.SimpleNamespace(decoder_embed_dim=CFG.embed_dim, decoder_ffn_embed_dim=CFG.ffn_dim, decoder_layers=CFG.layers,
                                 decoder_attention_heads=CFG.heads, decoder_normalize_before=True, decoder_learned_pos=False,
                                 encoder_embed_dim=CFG.embed_dim, encoder_ffn_embed_dim=CFG.ffn_dim, encoder_layers=2, encoder_attention_heads=CFG.heads,
                                 dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, activation_fn="relu", n_frames_per_step=1,
                                 max_target_positions=CFG.max_positions, share_decoder_input_output_embed=False, decoder_output_dim=CFG.embed_dim,
                                 decoder_input_dim=CFG.embed_dim, no_token_positional_embeddings=False, no_scale_embedding=False,
                                 layernorm_embedding=False, length_loss_factor=0.1)
    d = Dict1004()
    dec = R.nar.NARS2UTTransformerModel.build_decoder(args, d)
    sd = N.make_nar_state_dict(CFG, "nar")
    missing, unexpected = dec.load_state_dict(sd, strict=False)  # every PARAMETER is given; the two buffers are not weights
    assert set(missing) == {"version", "embed_positions._float_tensor"} and not unexpected, (missing, unexpected)
    assert {n for n, _ in dec.named_parameters()} == set(sd), "parameter names differ from nar_oracle.make_nar_state_dict"
    dec.eval()
    model = object.__new__(R.nar.NARS2UTTransformerModel)
    nn.Module.__init__(model)
    model.decoder, model.encoder = dec, N._Encoder()
    model.unk, model.pad, model.bos, model.eos = d.unk(), d.pad(), d.bos(), d.eos()
    model.allow_length_beam = True
    return model, d


def main():
    R = ref_loader.load_reference_nar()
    model, d = build_reference_model(R)
    out = {}
    B, S = 4, 23
    src_lens = torch.tensor([23, 9, 17, 14])
    enc = encoder_out(B, S, src_lens, 801)
    out["enc_out"], out["src_lens"] = enc["encoder_out"][0], src_lens
    # one decoder pass on a ragged, partially masked batch
    g = torch.Generator().manual_seed(802)
    tgt_lens = torch.tensor([31, 12, 2, 20])
    T = int(tgt_lens.max())
    tok = torch.randint(4, CFG.vocab, (B, T), generator=g)
    tok = torch.where(torch.rand(B, T, generator=g) < 0.5, torch.full_like(tok, d.unk()), tok)
    tok = tok.masked_fill(torch.arange(T)[None, :] >= tgt_lens[:, None], d.pad())
    with torch.no_grad():
        logits, _ = model.decoder(normalize=False, inference_mode=True, prev_output_tokens=tok, encoder_out=enc)
        lprobs, _ = model.decoder(normalize=True, inference_mode=True, prev_output_tokens=tok, encoder_out=enc)
        length_tgt = model.decoder.forward_length_prediction(model.decoder.forward_length(normalize=True, encoder_out=enc), encoder_out=enc)
    out.update(tokens=tok, logits=logits, lprobs_head=lprobs[:, :4], pred_lengths=length_tgt)
    # the research generator end to end (it hard-codes .to(device="cuda") for an index vector: mapped to the CPU here)
    orig_to = torch.Tensor.to

    def to_cpu(self, *a, **k):
        if k.get("device") == "cuda":
            k = dict(k, device="cpu")
        return orig_to(self, *a, **k)

    torch.Tensor.to = to_cpu
    try:
        for si, kw in enumerate(SETTINGS):
            gen = R.gen.IterativeRefinementGenerator(d, **kw)
            sample = {"net_input": {"src_tokens": torch.zeros(B, S, 80), "src_lengths": src_lens}}
            model.forward_encoder = lambda inputs, _e=enc: {k: list(v) for k, v in _e.items()}
            with torch.no_grad():
                hyps = gen.generate([model], sample)
            out[f"s{si}_n"] = len(hyps)
            for i, h in enumerate(hyps):
                out[f"s{si}_h{i}_tokens"], out[f"s{si}_h{i}_scores"] = h[0]["tokens"], h[0]["positional_scores"]
                out[f"s{si}_h{i}_steps"], out[f"s{si}_h{i}_score"] = h[0]["steps"], h[0]["score"]
                if kw.get("retain_history"):
                    out[f"s{si}_h{i}_nhist"] = len(h[0]["history"])
                    for j, hh in enumerate(h[0]["history"]):
                        out[f"s{si}_h{i}_hist{j}"] = hh["tokens"]
    finally:
        torch.Tensor.to = orig_to
    save("nar_decoder", **out)


if __name__ == "__main__":
    main()
