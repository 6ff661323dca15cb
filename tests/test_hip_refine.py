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
