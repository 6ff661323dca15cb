"""TEST INFRASTRUCTURE ONLY.  Golden vectors for the optimizer step, produced in the build container by the REAL
reference code: fairseq/optim/adam.py (class Adam, :97-239), fairseq/utils.py (clip_grad_norm_, :347-397) and
fairseq/optim/lr_scheduler/inverse_square_root_schedule.py (:31-85), driven the way fairseq's trainer drives them
(clip, then lr = scheduler.step_update(num_updates), then optimizer.step) with the recipe of scripts/diffusion/train.sh:29-31
(betas (0.9, 0.98), clip-norm 2.0, inverse_sqrt with warmup_init_lr 1e-7).  Seeded parameters and gradients of three
tensors; recorded after every update: parameters, exp_avg, exp_avg_sq, the gradient norm and the learning rate.
Usage:  python oracle/gen_golden_optim.py    ->  tests/golden/optim.npz
"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_loader  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "optim.npz")
SHAPES = [(37, 19), (1001,), (8, 3, 5)]
STEPS, LR, WARMUP, WARMUP_INIT, BETAS, EPS, CLIP = 6, 5e-4, 4, 1e-7, (0.9, 0.98), 1e-8, 2.0


def main():
    Adam, clip_grad_norm_, InverseSquareRootSchedule = ref_loader.load_reference_optim()
    rng = np.random.RandomState(7)
    out = {"meta": np.array([STEPS, WARMUP], dtype=np.int64), "hyper": np.array([LR, WARMUP_INIT, BETAS[0], BETAS[1], EPS, CLIP])}
    for wd_name, wd in (("", 0.0), ("wd_", 0.01)):
        params = [torch.nn.Parameter(torch.from_numpy(rng.randn(*s).astype(np.float32))) for s in SHAPES]
        opt = Adam(params, lr=LR, betas=BETAS, eps=EPS, weight_decay=wd)
        holder = types.SimpleNamespace(lr=None)
        holder.set_lr = lambda lr: [g.__setitem__("lr", lr) for g in opt.param_groups] and setattr(holder, "lr", lr)
        holder.get_lr = lambda: opt.param_groups[0]["lr"]
        cfg = types.SimpleNamespace(lr=[LR], warmup_updates=WARMUP, warmup_init_lr=WARMUP_INIT)
        sched = InverseSquareRootSchedule(cfg, holder)
        out[wd_name + "p0"] = np.concatenate([p.detach().numpy().ravel() for p in params])
        for it in range(STEPS):
            # gradient scale grows so that some updates clip (norm > 2) and some do not
            scale = 0.002 * (4.0 ** it)
            grads = [rng.randn(*s).astype(np.float32) * np.float32(scale) for s in SHAPES]
            out[f"{wd_name}g{it}"] = np.concatenate([g.ravel() for g in grads])
            for p, g in zip(params, grads):
                p.grad = torch.from_numpy(g.copy())
            norm = clip_grad_norm_(params, CLIP)
            lr = sched.step_update(it)  # the trainer sets the lr of update `it` from num_updates = it before stepping
            opt.step()
            out[f"{wd_name}norm{it}"] = np.float32(norm.item())
            out[f"{wd_name}lr{it}"] = np.float64(lr)
            out[f"{wd_name}p{it + 1}"] = np.concatenate([p.detach().numpy().ravel() for p in params])
            out[f"{wd_name}m{it + 1}"] = np.concatenate([opt.state[p]["exp_avg"].numpy().ravel() for p in params])
            out[f"{wd_name}v{it + 1}"] = np.concatenate([opt.state[p]["exp_avg_sq"].numpy().ravel() for p in params])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: (v.shape if hasattr(v, "shape") else v) for k, v in list(out.items())[:6]})


if __name__ == "__main__":
    main()
