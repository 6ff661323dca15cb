"""Optimizer step of the reference's training recipe on the HIP path (first piece): fairseq's Adam
(fairseq/optim/adam.py:97-239), gradient clipping by global norm (fairseq/utils.py:347-397) and the inverse_sqrt schedule
(fairseq/optim/lr_scheduler/inverse_square_root_schedule.py:31-85) -- scripts/diffusion/train.sh:29-31.  Parameters, gradients
and both moments live in flat fp32 device buffers (one launch each for the norm and the update, no host round trip between
them); the backward kernels that would fill the gradient buffer are not part of this round.
"""
import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _flat_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 1 and t.is_contiguous()):
        raise ValueError(f"{name}: expected a flat contiguous fp32 CUDA tensor")
    return t


class InverseSquareRootSchedule:
    """fairseq's `inverse_sqrt` (same arguments and `step_update` contract): linear warm-up from warmup_init_lr to lr over
    warmup_updates, then lr * sqrt(warmup_updates / num_updates)."""

    def __init__(self, lr: float, warmup_updates: int, warmup_init_lr: float = -1.0):
        if warmup_init_lr < 0:
            warmup_init_lr = 0 if warmup_updates > 0 else lr
        self.warmup_updates, self.warmup_init_lr = warmup_updates, warmup_init_lr
        self.lr_step = (lr - warmup_init_lr) / warmup_updates
        self.decay_factor = lr * warmup_updates ** 0.5
        self.lr = warmup_init_lr

    def step_update(self, num_updates: int) -> float:
        if num_updates < self.warmup_updates:
            self.lr = self.warmup_init_lr + num_updates * self.lr_step
        else:
            self.lr = self.decay_factor * num_updates ** -0.5
        return self.lr


class Adam:
    """fairseq.optim.adam.Adam over one flat fp32 parameter buffer (updated in place), with --clip-norm folded in.

    step(grad) = clip_grad_norm_(params, clip_norm) followed by optimizer.step(): returns the gradient norm (a device
    scalar, like the reference) without synchronising.  `bf16_copy` (same length, bf16) receives the updated parameters in
    the operand type of the forward kernels."""

    def __init__(self, params: torch.Tensor, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, clip_norm: float = 0.0, bf16_copy: Optional[torch.Tensor] = None):
        self.params = _flat_f32(params, "params")
        self.lr, self.betas, self.eps, self.weight_decay, self.clip_norm = lr, betas, eps, weight_decay, clip_norm
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.step_count = 0
        self.bf16_copy = bf16_copy
        if bf16_copy is not None and not (bf16_copy.is_cuda and bf16_copy.dtype == torch.bfloat16 and bf16_copy.numel() == params.numel()):
            raise ValueError("bf16_copy: expected a bf16 CUDA tensor of the parameters' length")
        self._scratch = torch.empty(1024 + 1, device=params.device, dtype=torch.float32)  # partial sums + the sum of squares

    def set_lr(self, lr: float):
        self.lr = lr

    def get_lr(self) -> float:
        return self.lr

    def grad_sumsq(self, grad: torch.Tensor, accumulate: bool = False) -> torch.Tensor:
        """Sum of squares of a gradient buffer into the optimizer's norm slot (accumulate=True adds a further buffer)."""
        lib = _lib.load()
        g = _flat_f32(grad, "grad")
        _lib.check(lib.dn_grad_sumsq(g.data_ptr(), g.numel(), self._scratch.data_ptr(), self._scratch[1024:].data_ptr(),
                                     int(accumulate), _stream()), "dn_grad_sumsq")
        return self._scratch[1024:]

    def step(self, grad: torch.Tensor, grad_scale: float = 1.0, grad_scale_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`grad_scale` (host) and `grad_scale_dev` (device scalar) are the trainer's multiply_grads factor (fairseq/trainer.py:
        918-933), applied before the norm and the clip; returns the norm of the SCALED gradient (a device scalar)."""
        lib = _lib.load()
        g = _flat_f32(grad, "grad")
        if g.numel() != self.params.numel():
            raise ValueError("grad and params differ in length")
        sumsq = self.grad_sumsq(g)
        self.step_count += 1
        hp = _lib.AdamParams(lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, weight_decay=self.weight_decay,
                             max_norm=self.clip_norm, step=self.step_count, grad_scale=grad_scale,
                             grad_scale_dev=_lib.ptr(grad_scale_dev))
        _lib.check(lib.dn_adam_step(self.params.data_ptr(), g.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                    g.numel(), C.byref(hp), sumsq.data_ptr(), _lib.ptr(self.bf16_copy), _stream()), "dn_adam_step")
        norm = sumsq.sqrt()[0] * grad_scale
        return norm * grad_scale_dev.reshape(-1)[0] if grad_scale_dev is not None else norm


class FlatOptimizer:
    """The part of fairseq's FairseqOptimizer contract a training step uses (fairseq/optim/fairseq_optimizer.py: backward,
    multiply_grads, clip_grad_norm, step, zero_grad, set_lr / get_lr, state_dict) over the flat buffers of a HIP training
    engine (diffnorm_amd/training.py).  multiply_grads and the clip coefficient are not separate passes over the gradient:
    they are recorded and applied inside the Adam kernel (fairseq's own note at trainer.py:925-927 allows exactly this)."""

    def __init__(self, engine, lr: float = 5e-4, betas=(0.9, 0.98), eps: float = 1e-8, weight_decay: float = 0.0):
        self.engine = engine
        self.adam = Adam(engine.master, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, clip_norm=0.0,
                         bf16_copy=engine.work if engine.work is not engine.master else None)
        self._scale, self._scale_dev, self._max_norm = 1.0, None, 0.0

    def backward(self, loss):
        loss.backward()

    def multiply_grads(self, c):
        if torch.is_tensor(c):
            c = c.to(self.engine.device, torch.float32).reshape(1)
            self._scale_dev = c if self._scale_dev is None else self._scale_dev * c
        else:
            self._scale *= float(c)

    def clip_grad_norm(self, max_norm, aggregate_norm_fn=None):
        """Returns the norm of the (scaled) gradient as a device scalar; the clip itself happens inside `step`."""
        from .profiling import profile_range

        self._max_norm = float(max_norm)
        with profile_range("clip-grads"):  # fairseq/trainer.py:937 (here: the norm; the clip coefficient is applied by dn_adam_step)
            norm = self.adam.grad_sumsq(self.engine.grads).sqrt()[0] * self._scale
        return norm * self._scale_dev[0] if self._scale_dev is not None else norm

    def step(self, closure=None):
        from .profiling import profile_range

        self.adam.clip_norm = self._max_norm
        with profile_range("optimizer"):  # fairseq/trainer.py:958
            self.adam.step(self.engine.grads, grad_scale=self._scale, grad_scale_dev=self._scale_dev)
            self.engine.refresh()
        self._scale, self._scale_dev = 1.0, None

    def zero_grad(self):
        self.engine.zero_grad()
        self._scale, self._scale_dev = 1.0, None

    def set_lr(self, lr):
        self.adam.set_lr(lr)

    def get_lr(self):
        return self.adam.get_lr()

    def state_dict(self):
        """Adam moments in the reference's parameter layout would need the per-tensor optimizer state of fairseq's Adam; the flat
        moments are stored as they are, with the step count (resuming needs the same packed layout)."""
        return {"step": self.adam.step_count, "exp_avg": self.adam.exp_avg.detach().cpu(), "exp_avg_sq": self.adam.exp_avg_sq.detach().cpu(),
                "lr": self.adam.get_lr()}

    def load_state_dict(self, sd):
        self.adam.step_count = int(sd["step"])
        self.adam.exp_avg.copy_(sd["exp_avg"])
        self.adam.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.adam.set_lr(float(sd["lr"]))
