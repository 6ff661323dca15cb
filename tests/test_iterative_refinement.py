"""SURVEY 8 f4, host logic: diffnorm_amd.iterative_refinement.IterativeRefinementGenerator against hypotheses the REAL reference
generator produced (oracle/gen_golden_refine.py -> tests/golden/refine.npz: fairseq/iterative_refinement_generator.py driving the
toy NAT model of oracle/toy_nat.py with the reference's own _skeptical_unmasking), for adaptive / fixed iteration counts, history,
a length beam of 3 and true-length decoding; and the restated mask-predict update against the reference's.  CPU only."""
import numpy as np
import torch

import toy_nat


def check_hypos(hypos, g, k):
    assert len(hypos) == int(g[f"s{k}_n"])
    for i, h in enumerate(hypos):
        h = h[0]
        assert h["tokens"].cpu().tolist() == g[f"s{k}_{i}_tokens"].tolist(), (k, i)
        assert int(h["steps"]) == int(g[f"s{k}_{i}_steps"]), (k, i)
        np.testing.assert_allclose(h["positional_scores"].cpu().numpy(), g[f"s{k}_{i}_scores"], rtol=2e-4, atol=2e-5)
        assert abs(float(h["score"]) - float(g[f"s{k}_{i}_scores"].mean())) < 1e-4
        if f"s{k}_{i}_nhist" in g:
            assert len(h["history"]) == int(g[f"s{k}_{i}_nhist"])
            for j, hh in enumerate(h["history"]):
                assert hh["tokens"].cpu().tolist() == g[f"s{k}_{i}_hist{j}"].tolist(), (k, i, j)


def test_generator_matches_the_reference_generator(golden):
    from diffnorm_amd.iterative_refinement import IterativeRefinementGenerator

    g = golden("refine")
    d = toy_nat.ToyDict()
    for k, kw in enumerate(toy_nat.SETTINGS):
        model = toy_nat.ToyCMLM(d, toy_nat.torch_update())
        gen = IterativeRefinementGenerator(d, **kw)
        check_hypos(gen.generate([model], toy_nat.toy_sample(d)), g, k)
    # the dataset iterator strips padding from source and reference (:63-102)
    gen = IterativeRefinementGenerator(d, models=[toy_nat.ToyCMLM(d, toy_nat.torch_update())], max_iter=4)
    rows = list(gen.generate_batched_itr([toy_nat.toy_sample(d), {"no_net_input": 1}]))
    assert len(rows) == 5 and rows[1][1].numel() == 3 and rows[1][2].numel() == 5
    assert rows[2][3][0]["tokens"].tolist() == g["s0_2_tokens"].tolist()


def test_mask_predict_update_restatement(golden):
    g = golden("refine")
    gen = torch.Generator().manual_seed(int(g["u_logits_seed"]))
    logits = torch.randn(4, 37, 1004, generator=gen) * 3
    upd = toy_nat.torch_update()
    for c in range(4):
        step, max_step = (int(v) for v in g[f"u{c}_step"])
        pred, tok, sc = upd(logits, torch.from_numpy(g[f"u{c}_tok_in"]), torch.from_numpy(g[f"u{c}_sc_in"]), step, max_step)
        assert pred.tolist() == g[f"u{c}_pred"].tolist() and tok.tolist() == g[f"u{c}_tok_out"].tolist()
        np.testing.assert_allclose(sc.numpy(), g[f"u{c}_sc_out"], rtol=1e-5, atol=1e-6)
