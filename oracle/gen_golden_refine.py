"""TEST INFRASTRUCTURE ONLY.  Golden vectors for SURVEY 8 f4 from the REAL reference code: the class
fairseq/iterative_refinement_generator.py::IterativeRefinementGenerator (leaf-loaded; it only needs fairseq.utils) driving the toy
NAT model of oracle/toy_nat.py whose mask-predict update calls the reference's own `_skeptical_unmasking`
(fairseq/models/nat/cmlm_transformer.py:19-25; that file imports the whole NAT model zoo, so the one function is compiled from its
source text in place).  Stored: per setting and sentence the hypothesis tokens, positional scores, finishing step and history, and
stand-alone update cases (logits, tokens/scores before and after).   Usage: python oracle/gen_golden_refine.py"""
import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402
import toy_nat  # noqa: E402
from gen_golden import save  # noqa: E402


def reference_unmasking():
    ref_loader.load_reference()
    path = os.path.join(ref_loader.REF, "fairseq/models/nat/cmlm_transformer.py")
    tree = ast.parse(open(path).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "_skeptical_unmasking")
    ns = {"new_arange": sys.modules["fairseq.utils"].new_arange}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    return ns["_skeptical_unmasking"]


def main():
    unmask = reference_unmasking()
    gen_mod = ref_loader._load("fairseq.iterative_refinement_generator", "fairseq/iterative_refinement_generator.py")
    d = toy_nat.ToyDict()
    out = {}
    for k, kw in enumerate(toy_nat.SETTINGS):
        model = toy_nat.ToyCMLM(d, toy_nat.torch_update(unmask))
        gen = gen_mod.IterativeRefinementGenerator(d, **kw)
        hypos = gen.generate([model], toy_nat.toy_sample(d))
        out[f"s{k}_n"] = len(hypos)
        for i, h in enumerate(hypos):
            h = h[0]
            out[f"s{k}_{i}_tokens"] = h["tokens"]
            out[f"s{k}_{i}_scores"] = h["positional_scores"]
            out[f"s{k}_{i}_steps"] = h["steps"]
            if "history" in h:
                out[f"s{k}_{i}_nhist"] = len(h["history"])
                for j, hh in enumerate(h["history"]):
                    out[f"s{k}_{i}_hist{j}"] = hh["tokens"]
    # stand-alone update cases: [B, T, V] logits, ragged rows, several (step, max_step)
    g = torch.Generator().manual_seed(9)
    B, T, V = 4, 37, 1004
    logits = torch.randn(B, T, V, generator=g) * 3
    lens = torch.tensor([37, 20, 2, 9])
    for c, (step, max_step) in enumerate(((0, 5), (2, 5), (4, 5), (0, 11))):
        tok = torch.full((B, T), d.pad(), dtype=torch.long)
        sc = torch.zeros(B, T)
        for b, n in enumerate(lens):
            tok[b, :n] = d.unk()
            tok[b, 0], tok[b, n - 1] = d.bos(), d.eos()
            known = torch.rand(int(n), generator=g) < 0.4  # some positions already decided in earlier iterations
            known[0] = known[n - 1] = False
            tok[b, :n][known] = torch.randint(4, V, (int(known.sum()),), generator=g)
            sc[b, :n][known] = -torch.rand(int(known.sum()), generator=g) * 3
        pred, tok2, sc2 = toy_nat.torch_update(unmask)(logits, tok, sc, step, max_step)
        out.update({f"u{c}_tok_in": tok, f"u{c}_sc_in": sc, f"u{c}_pred": pred, f"u{c}_tok_out": tok2, f"u{c}_sc_out": sc2,
                    f"u{c}_step": np.array([step, max_step])})
    out["u_logits_seed"] = 9
    save("refine", **out)


if __name__ == "__main__":
    main()
