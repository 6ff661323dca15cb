"""Per-tensor gradient agreement of the bf16 VAE training step with the fp32 oracle (diagnostic; GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, torch
import diffnorm_oracle as O, train_oracle as TO
from gen_golden_configs import CHAIN_VAE as CFG, seeded
from diffnorm_amd import training

g = np.load(os.path.join(ROOT, "tests/golden/vae_train.npz"))
sd = O.make_vae_state_dict(CFG, "train")
feat, units, lens = seeded((3, 48, CFG.dim), 31), torch.from_numpy(g["units"]), torch.from_numpy(g["lens"])
noise = torch.from_numpy(g["post_noise"])
_, want = TO.vae_loss_and_grads(sd, CFG, feat, units, lens, noise)
tot = float(torch.sqrt(sum(v.double().pow(2).sum() for v in want.values())))
for dt in ("f32", "bf16"):
    eng = training.VaeTrainEngine(sd, dim=CFG.dim, latent_dim=CFG.latent_dim, dtype=dt, depth=CFG.depth, heads=CFG.heads,
                                  dim_head=CFG.dim_head, stacks=CFG.stacks, layers=CFG.layers)
    eng.forward(feat, units, lens, noise=noise); eng.zero_grad(); eng.backward()
    got = eng.grad_dict()
    rows = []
    for k in want:
        w, q = want[k].double(), got[k].double()
        rows.append((float((q - w).norm() / tot), float(w.norm() / tot), float((q * w).sum() / (q.norm() * w.norm() + 1e-30)), k))
    rows.sort(reverse=True)
    print(dt, "worst 12 by error / total norm:")
    for r in rows[:12]:
        print("  err/tot %.2e  |g|/tot %.2e  cos %.4f  %s" % r)
