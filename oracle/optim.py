"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (diffnorm_amd/).

CPU restatement (numpy, fp32 arithmetic) of the optimizer step of the reference's training recipe (SURVEY 8 f2;
scripts/diffusion/train.sh:29-31: --optimizer adam --adam-betas '(0.9,0.98)' --clip-norm 2.0 --lr-scheduler inverse_sqrt).
Pinned to the reference by tests/golden/optim.npz, which oracle/gen_golden_optim.py produces by running the reference's own
Adam class, clip_grad_norm_ and InverseSquareRootSchedule.
"""
import math

import numpy as np

F = np.float32


def total_norm(grads):
    """fairseq/utils.py:367-387: the 2-norm of the per-tensor 2-norms (fp32)."""
    norms = np.array([np.sqrt(np.sum(g.astype(F) * g.astype(F), dtype=F)) for g in grads], dtype=F)
    return F(norms[0]) if len(norms) == 1 else F(np.sqrt(np.sum(norms * norms, dtype=F)))


def clip_coef(norm, max_norm):
    """fairseq/utils.py:392-394: (max_norm / (total_norm + 1e-6)).clamp_(max=1); no clipping for max_norm <= 0."""
    if max_norm <= 0:
        return F(1.0)
    return min(F(F(max_norm) / (F(norm) + F(1e-6))), F(1.0))


def adam_step(p, g, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """fairseq/optim/adam.py:207-239 on fp32 arrays; returns the new (p, exp_avg, exp_avg_sq).  step: 1 for the first update."""
    b1, b2 = betas
    g = g.astype(F)
    m = m.astype(F) * F(b1) + F(1 - b1) * g                       # :215
    v = v.astype(F) * F(b2) + F(1 - b2) * (g * g)                 # :216
    denom = np.sqrt(v) + F(eps)                                   # :223
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step                     # :225-226
    step_size = lr * math.sqrt(bc2) / bc1                         # :227
    p = p.astype(F)
    if weight_decay != 0:
        p = p + p * F(-weight_decay * lr)                         # :229-232
    p = p + F(-step_size) * (m / denom)                           # :234
    return p.astype(F), m.astype(F), v.astype(F)


def inverse_sqrt_lr(num_updates, lr, warmup_updates, warmup_init_lr=-1.0):
    """fairseq/optim/lr_scheduler/inverse_square_root_schedule.py:58-69 (constructor), :78-85 (step_update)."""
    if warmup_init_lr < 0:
        warmup_init_lr = 0 if warmup_updates > 0 else lr
    lr_step = (lr - warmup_init_lr) / warmup_updates
    decay_factor = lr * warmup_updates ** 0.5
    if num_updates < warmup_updates:
        return warmup_init_lr + num_updates * lr_step
    return decay_factor * num_updates ** -0.5
