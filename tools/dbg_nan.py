import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import diffnorm_oracle as O
from gen_golden_configs import TINY_EPS
from diffnorm_amd import engine
g = dict(np.load(os.path.join(ROOT, "tests/golden/eps_tiny.npz")))
sd = O.make_eps_state_dict(TINY_EPS, "tiny")
x, t, lens = [torch.from_numpy(g[k]) for k in ("x", "t", "lens")]
for dtype in ("f32", "bf16"):
    for shared in (False, True):
        e = engine.EpsEngine(sd, TINY_EPS, dtype=dtype, device="cuda:0")
        tt = t if not shared else torch.full_like(t, 3)
        out = e.forward(x.cuda(), tt, lens, shared_t=shared).cpu()
        nan = torch.isnan(out)
        print(dtype, "shared" if shared else "per-sample", "nan count", int(nan.sum()), "per sample", nan.flatten(1).sum(1).tolist(),
              "first nan pos", (nan.nonzero()[0].tolist() if nan.any() else None))
