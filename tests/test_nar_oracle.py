"""CPU: the oracle's restatement of the NAR S2UT decoder (oracle/nar_oracle.py) against the REAL reference classes' outputs
(tests/golden/nar_decoder.npz: fairseq's TransformerDecoder stack + research/TranSpeech/nar_transformer.py + the research
IterativeRefinementGenerator, oracle/gen_golden_nar.py), and the generator mirror (speech_source=True) reproducing every
hypothesis of the reference generator when it drives the oracle model."""
import numpy as np
import torch

import nar_oracle as N
from gen_golden_nar_configs import CFG, SETTINGS, Dict1004, encoder_out


def T_(a):
    return torch.from_numpy(np.asarray(a))


def test_oracle_decoder_pass_and_length_prediction(golden):
    g = golden("nar_decoder")
    sd = N.make_nar_state_dict(CFG, "nar")
    enc, lens = T_(g["enc_out"]), T_(g["src_lens"])
    pad = torch.arange(enc.shape[0])[None, :] >= lens[:, None]
    with torch.no_grad():
        logits = N.decoder_logits(sd, CFG, T_(g["tokens"]), enc, pad, normalize=False)
        lp = N.decoder_logits(sd, CFG, T_(g["tokens"]), enc, pad, normalize=True)
    valid = T_(g["tokens"]).ne(CFG.pad)
    assert (logits - T_(g["logits"]))[valid].abs().max().item() < 5e-5
    assert (lp[:, :4] - T_(g["lprobs_head"])).abs().max().item() < 5e-5
    assert torch.equal(N.predict_lengths(sd, enc, pad), T_(g["pred_lengths"]))


def check_hypotheses(g, make_generator, model, sample, tol=2e-4):
    for si, kw in enumerate(SETTINGS):
        hyps = make_generator(kw).generate([model], sample)
        assert len(hyps) == int(g[f"s{si}_n"])
        for i, h in enumerate(hyps):
            assert h[0]["tokens"].cpu().tolist() == g[f"s{si}_h{i}_tokens"].tolist(), (si, i)
            assert int(h[0]["steps"]) == int(g[f"s{si}_h{i}_steps"])
            assert np.abs(h[0]["positional_scores"].cpu().numpy() - g[f"s{si}_h{i}_scores"]).max() < tol
            if kw.get("retain_history"):
                assert len(h[0]["history"]) == int(g[f"s{si}_h{i}_nhist"])
                for j, hh in enumerate(h[0]["history"]):
                    assert hh["tokens"].cpu().tolist() == g[f"s{si}_h{i}_hist{j}"].tolist(), (si, i, j)


def test_generator_mirror_reproduces_the_reference_hypotheses_with_the_oracle_model(golden):
    from diffnorm_amd.iterative_refinement import IterativeRefinementGenerator

    g = golden("nar_decoder")
    model = N.OracleNarModel(N.make_nar_state_dict(CFG, "nar"), CFG)
    lens = T_(g["src_lens"])
    enc = encoder_out(lens.numel(), g["enc_out"].shape[0], lens, 801)
    assert torch.equal(enc["encoder_out"][0], T_(g["enc_out"]))
    model.forward_encoder = lambda inputs: {k: list(v) for k, v in enc.items()}
    sample = {"net_input": {"src_tokens": torch.zeros(lens.numel(), g["enc_out"].shape[0], 80), "src_lengths": lens}}
    with torch.no_grad():
        check_hypotheses(g, lambda kw: IterativeRefinementGenerator(Dict1004(), speech_source=True, **kw), model, sample)


def _encoder_inputs(g):
    from gen_golden_configs import seeded
    from gen_golden_nar_configs import ENC_CFG

    lens = T_(g["lens"])
    B, L = lens.numel(), int(lens.max())
    feats = seeded((B, L, ENC_CFG.input_dim), 901) * (torch.arange(L)[None, :, None] < lens[:, None, None])
    return feats, lens


def test_oracle_speech_encoder_matches_the_real_reference(golden):
    """The speech encoder of the NAR S2UT model (round 4): oracle/nar_oracle.encoder_forward against the REAL S2STransformerEncoder over
    fairseq's S2TTransformerEncoder (tests/golden/nar_encoder.npz, oracle/gen_golden_nar_encoder.py) on a ragged batch: encoder output,
    padding mask, subsampled lengths."""
    from gen_golden_nar_configs import ENC_CFG

    g = golden("nar_encoder")
    feats, lens = _encoder_inputs(g)
    with torch.no_grad():
        eo, pad, ol = N.encoder_forward(N.make_nar_encoder_state_dict(ENC_CFG, "narenc"), ENC_CFG, feats, lens)
    assert (eo - T_(g["encoder_out"])).abs().max().item() < 2e-5
    assert torch.equal(pad, T_(g["padding_mask"])) and ol.tolist() == g["out_lens"].tolist()
