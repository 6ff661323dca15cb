"""Timestep respacing (reference diffusion/respace.py:12-129)."""
import numpy as np
import torch

from .gaussian_diffusion import GaussianDiffusion


def space_timesteps(num_timesteps, section_counts):
    """Set of original steps to keep: "ddimN" = fixed stride giving exactly N steps, otherwise comma-separated counts per
    equal section with fractional striding inside each (reference :12-62)."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per, extra = divmod(num_timesteps, len(section_counts))
    start, steps = 0, []
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            steps.append(start + round(cur))
            cur += stride
        start += size
    return set(steps)


class _WrappedModel:
    """Maps respaced step indices back to the original ones before calling the model (reference :117-129)."""

    def __init__(self, model, timestep_map, original_num_steps):
        self.model, self.timestep_map, self.original_num_steps = model, timestep_map, original_num_steps

    def __call__(self, x, ts, **kwargs):
        m = torch.tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)
        return self.model(x, m[ts], **kwargs)


class SpacedDiffusion(GaussianDiffusion):
    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.timestep_map = []
        self.original_num_steps = len(kwargs["betas"])
        base = GaussianDiffusion(**kwargs)
        last, new_betas = 1.0, []
        for i, ac in enumerate(base.alphas_cumprod):
            if i in self.use_timesteps:
                new_betas.append(1 - ac / last)
                last = ac
                self.timestep_map.append(i)
        kwargs["betas"] = np.array(new_betas)
        super().__init__(**kwargs)

    def _wrap_model(self, model):
        return model if isinstance(model, _WrappedModel) else _WrappedModel(model, self.timestep_map, self.original_num_steps)

    def _step(self, model, *args, cond_fn=None, **kwargs):  # the samplers: model and cond_fn both see the ORIGINAL step indices (:102-106)
        return super()._step(self._wrap_model(model), *args, cond_fn=None if cond_fn is None else self._wrap_model(cond_fn), **kwargs)

    def condition_mean(self, cond_fn, *args, **kwargs):
        return super().condition_mean(self._wrap_model(cond_fn), *args, **kwargs)

    def condition_score(self, cond_fn, *args, **kwargs):
        return super().condition_score(self._wrap_model(cond_fn), *args, **kwargs)

    def _model_out(self, model, *args, **kwargs):  # p_mean_variance / ddim_reverse_sample / _vb_terms_bpd (reference :90-93)
        return super()._model_out(self._wrap_model(model), *args, **kwargs)

    def training_losses(self, model, *args, **kwargs):
        return super().training_losses(self._wrap_model(model), *args, **kwargs)
