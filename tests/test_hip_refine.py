"""SURVEY 8 f4 on the GPU: the mask-predict update kernel (dn_cmlm_step) against the reference's update
(tests/golden/refine.npz), and the whole refinement loop with the kernel inside the toy model's forward_decoder against the
hypotheses of the real reference generator."""
import numpy as np
import pytest
import torch

import toy_nat
from test_iterative_refinement import check_hypos

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def hip_update(logits, tokens, scores, step, max_step):
    from diffnorm_amd.iterative_refinement import cmlm_update

    tok = tokens.to(torch.int32).contiguous().clone()
    sc = scores.float().contiguous().clone()
    pred, tok = cmlm_update(logits.float().contiguous(), tok, sc, step, max_step, unk=3, pad=1)
    return pred.long(), tok.long(), sc


def test_cmlm_update_kernel(golden):
    g = golden("refine")
    gen = torch.Generator().manual_seed(int(g["u_logits_seed"]))
    logits = (torch.randn(4, 37, 1004, generator=gen) * 3).to(DEV)
    for c in range(4):
        step, max_step = (int(v) for v in g[f"u{c}_step"])
        pred, tok, sc = hip_update(logits, torch.from_numpy(g[f"u{c}_tok_in"]).to(DEV), torch.from_numpy(g[f"u{c}_sc_in"]).to(DEV), step, max_step)
        assert pred.cpu().tolist() == g[f"u{c}_pred"].tolist(), c
        assert tok.cpu().tolist() == g[f"u{c}_tok_out"].tolist(), c
        np.testing.assert_allclose(sc.cpu().numpy(), g[f"u{c}_sc_out"], rtol=1e-5, atol=2e-6)


def test_refinement_loop_with_the_kernel(golden):
    from diffnorm_amd.iterative_refinement import IterativeRefinementGenerator

    g = golden("refine")
    d = toy_nat.ToyDict()
    for k, kw in enumerate(toy_nat.SETTINGS):
        model = toy_nat.ToyCMLM(d, hip_update, device=DEV)
        check_hypos(IterativeRefinementGenerator(d, **kw).generate([model], toy_nat.toy_sample(d, DEV)), g, k)


# ---------------------------------------------------------------------------------------------------------------- the model inside the loop
@pytest.mark.parametrize("dtype,tol", [("f32", 1e-3), ("bf16x3", 1e-3), ("f16", 1e-2), ("bf16", 6e-2)])
def test_nar_decoder_pass_matches_the_real_reference(golden, dtype, tol):
    """SURVEY 8 f4, the model inside the mask-predict loop: one pass of the NAR S2UT decoder on the HIP engine (dn_nar_decoder_forward:
    embedding + positions, pre-norm self-attention / encoder attention / ReLU FFN layers, final LayerNorm, 1004-way projection) and
    the length predictor against the REAL reference classes' outputs (tests/golden/nar_decoder.npz: fairseq's TransformerDecoder
    stack under research/TranSpeech/nar_transformer.py's TransformerUnitDecoder) on a ragged, partially masked batch."""
    import nar_oracle as N
    from diffnorm_amd import nar_decoder
    from gen_golden_nar_configs import CFG

    g = golden("nar_decoder")
    eng = nar_decoder.NarDecoderEngine(N.make_nar_state_dict(CFG, "nar"), CFG.embed_dim, CFG.ffn_dim, CFG.layers, CFG.heads, CFG.vocab, dtype=dtype, device=DEV)
    enc = torch.from_numpy(g["enc_out"]).transpose(0, 1).contiguous().to(DEV)  # [S,B,D] -> [B,S,D]
    slen = torch.from_numpy(g["src_lens"]).to(DEV).int()
    tok = torch.from_numpy(g["tokens"])
    ckv = eng.cross_kv(enc)
    logits = eng.forward(tok.to(DEV).int().contiguous(), ckv, slen).cpu()
    valid = tok.ne(CFG.pad)
    err = (logits - torch.from_numpy(g["logits"]))[valid].abs().max().item()
    print(f"NAR decoder logits {dtype}: max abs err {err:.3e} (logit scale {np.abs(g['logits']).max():.2f})")
    assert err < tol * max(1.0, float(np.abs(g["logits"]).max()) / 4)
    assert eng.predict_lengths(enc, slen).cpu().tolist() == g["pred_lengths"].tolist()


@pytest.mark.parametrize("dtype", ["f32", "bf16x3"])
def test_hip_nar_model_in_the_research_generator_reproduces_the_reference_hypotheses(golden, dtype):
    """The loop for real: the generator mirror in its research/TranSpeech flavour (3-D speech source, initialize_output_tokens(encoder_out,
    src_lengths)) driving NARS2UTDecoderModel -- HIP decoder passes + dn_cmlm_step -- reproduces every hypothesis (tokens, steps,
    history exactly; scores 2e-4) the REAL reference generator produced with the REAL reference model: adaptive early stop with a
    shrinking batch, a fixed 10 iterations with history, a length beam of 3, and max_iter = 0."""
    import nar_oracle as N
    from diffnorm_amd import nar_decoder
    from diffnorm_amd.iterative_refinement import IterativeRefinementGenerator
    from gen_golden_nar_configs import CFG, SETTINGS, Dict1004, encoder_out
    from test_nar_oracle import check_hypotheses

    g = golden("nar_decoder")
    model = nar_decoder.NARS2UTDecoderModel(N.make_nar_state_dict(CFG, "nar"), CFG.embed_dim, CFG.ffn_dim, CFG.layers, CFG.heads, CFG.vocab, dtype=dtype,
                                            device=DEV)
    lens = torch.from_numpy(g["src_lens"])
    enc = encoder_out(lens.numel(), g["enc_out"].shape[0], lens, 801)
    to_dev = lambda e: {k: [x.to(DEV) for x in v] for k, v in e.items()}
    model.forward_encoder = lambda inputs: to_dev(enc)
    sample = {"net_input": {"src_tokens": torch.zeros(lens.numel(), g["enc_out"].shape[0], 80, device=DEV), "src_lengths": lens.to(DEV)}}
    check_hypotheses(g, lambda kw: IterativeRefinementGenerator(Dict1004(), speech_source=True, **kw), model, sample)


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-3), ("bf16x3", 1e-3), ("f16", 1e-2), ("bf16", 6e-2)])
def test_nar_speech_encoder_matches_the_real_reference(golden, dtype, tol):
    """The speech encoder in front of the loop (round 4; dn_nar_encoder_forward: Conv1dSubsampler as two gathered-row contractions with
    the GLUs folded into the gathers, sqrt(D) scaling + positions of the padding mask, pre-norm self-attention / ReLU FFN layers,
    LayerNorm) against the REAL S2STransformerEncoder's output (tests/golden/nar_encoder.npz) on a ragged batch of [B, L, 80] features."""
    import nar_oracle as N
    from diffnorm_amd import nar_decoder
    from gen_golden_nar_configs import ENC_CFG as E
    from test_nar_oracle import _encoder_inputs

    g = golden("nar_encoder")
    feats, lens = _encoder_inputs(g)
    eng = nar_decoder.NarEncoderEngine(N.make_nar_encoder_state_dict(E, "narenc"), E.input_dim, E.conv_channels, E.kernel_sizes[0], E.embed_dim, E.ffn_dim,
                                       E.layers, E.heads, dtype=dtype, device=DEV)
    out, ol = eng.forward(feats.to(DEV), lens.to(DEV))
    assert ol.cpu().tolist() == g["out_lens"].tolist()
    ref = torch.from_numpy(g["encoder_out"]).transpose(0, 1)  # [S, B, D] -> [B, S, D]
    valid = ~torch.from_numpy(g["padding_mask"])
    err = (out.cpu() - ref)[valid].abs().max().item()
    print(f"NAR speech encoder {dtype}: max abs err {err:.3e} (output scale {ref.abs().max().item():.2f})")
    assert err < tol * max(1.0, ref.abs().max().item() / 4)


@pytest.mark.parametrize("dtype", ["f32", "bf16x3"])
def test_generate_from_features_reproduces_the_reference(golden, dtype):
    """generate() of the research generator from [B, L, 80] fbank-like features, the whole model on the HIP engines -- speech encoder,
    length predictor, encoder-attention keys / values, the refinement iterations (each one captured hipGraph launch) -- reproduces the
    hypotheses the REAL generator produced with the REAL NARS2UTTransformerModel-style model (encoder + decoder of the fixtures)."""
    import nar_oracle as N
    from diffnorm_amd import nar_decoder
    from diffnorm_amd.iterative_refinement import IterativeRefinementGenerator
    from gen_golden_nar_configs import CFG, ENC_CFG as E, Dict1004
    from test_nar_oracle import _encoder_inputs

    g = golden("nar_encoder")
    feats, lens = _encoder_inputs(g)
    model = nar_decoder.NARS2UTDecoderModel(
        N.make_nar_state_dict(CFG, "nar"), CFG.embed_dim, CFG.ffn_dim, CFG.layers, CFG.heads, CFG.vocab, dtype=dtype, device=DEV,
        encoder_state_dict=N.make_nar_encoder_state_dict(E, "narenc"),
        encoder_kw=dict(input_dim=E.input_dim, conv_channels=E.conv_channels, kernel=E.kernel_sizes[0], ffn=E.ffn_dim, layers=E.layers))
    gen = IterativeRefinementGenerator(Dict1004(), speech_source=True, max_iter=4, beam_size=1, adaptive=True)
    hyps = gen.generate([model], {"net_input": {"src_tokens": feats.to(DEV), "src_lengths": lens.to(DEV)}})
    assert len(hyps) == int(g["gen_n"])
    for i, h in enumerate(hyps):
        assert h[0]["tokens"].cpu().tolist() == g[f"gen_h{i}_tokens"].tolist(), i
        assert int(h[0]["steps"]) == int(g[f"gen_h{i}_steps"])
        assert np.abs(h[0]["positional_scores"].cpu().numpy() - g[f"gen_h{i}_scores"]).max() < 2e-4
