"""Noise schedules of the path: float64 host tables (as the reference keeps them) and the fp32
coefficient tables the HIP scheduler kernels gather from.

Mirrors ``DDPMScheduler`` (reference latent_module.py:1241-1297) by name and argument meaning; the
tables are constants built once on the host in NumPy float64, exactly like upstream; every per-sample
use on the device goes through dn_q_sample / dn_ddim_step.
"""
import math

import numpy as np
import torch


def betas_for_alpha_bar(n: int, alpha_bar, max_beta: float = 0.999) -> np.ndarray:
    """reference latent_module.py:1145-1162."""
    return np.array([min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), max_beta) for i in range(n)])


def get_named_beta_schedule(name: str, n: int) -> np.ndarray:
    """reference latent_module.py:1199-1223 ("linear" | "cosine")."""
    if name == "linear":
        scale = 1000 / n
        return np.linspace(scale * 0.0001, scale * 0.02, n, dtype=np.float64)
    if name == "cosine":
        return betas_for_alpha_bar(n, lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {name}")


class ScheduleTables:
    """The float64 tables shared by DDPMScheduler and GaussianDiffusion (:1247-1276)."""

    def __init__(self, betas: np.ndarray):
        betas = np.asarray(betas, dtype=np.float64)
        assert betas.ndim == 1 and (betas > 0).all() and (betas <= 1).all()
        self.betas = betas
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = (
            np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
            if len(self.posterior_variance) > 1 else np.array([]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)

    def f32(self, name_or_array, device=None) -> torch.Tensor:
        """fp32 cast of a table, the form every gather upstream produces (`.float()`, :1235)."""
        if isinstance(name_or_array, str):  # the named tables are immutable: one device copy each (a pageable H2D copy per call
            key = (name_or_array, str(device))  # synchronises the stream -- per training update, that drained the pipeline)
            cache = self.__dict__.setdefault("_f32_cache", {})
            if key not in cache:
                t = torch.from_numpy(np.ascontiguousarray(getattr(self, name_or_array).astype(np.float32)))
                cache[key] = t.to(device) if device is not None else t
            return cache[key]
        t = torch.from_numpy(np.ascontiguousarray(name_or_array.astype(np.float32)))
        return t.to(device) if device is not None else t

    def gaussian_table(self, device=None, fixed_large: bool = False) -> torch.Tensor:
        """fp32 [timesteps, DN_GD_COLS = 12] table of dn_gaussian_step / dn_gaussian_moments / dn_ddpm_loop: {sqrt_recip_abar,
        sqrt_recipm1_abar, posterior_mean_coef1, posterior_mean_coef2, fixed log-variance (FIXED_SMALL: the clipped posterior
        log-variance; FIXED_LARGE: log [posterior_variance[1], betas[1:]], diffusion/gaussian_diffusion.py:300-310),
        posterior_log_variance_clipped, log betas, abar, abar_prev, abar_next, posterior_variance, the fixed variance}."""
        fixed_var = np.append(self.posterior_variance[1], self.betas[1:]) if fixed_large else self.posterior_variance
        fixed_log = np.log(np.append(self.posterior_variance[1], self.betas[1:])) if fixed_large else self.posterior_log_variance_clipped
        cols = [self.sqrt_recip_alphas_cumprod, self.sqrt_recipm1_alphas_cumprod, self.posterior_mean_coef1, self.posterior_mean_coef2,
                fixed_log, self.posterior_log_variance_clipped, np.log(self.betas), self.alphas_cumprod, self.alphas_cumprod_prev,
                np.append(self.alphas_cumprod[1:], 0.0), self.posterior_variance, fixed_var]
        t = torch.from_numpy(np.stack(cols, axis=1).astype(np.float32)).contiguous()
        return t.to(device) if device is not None else t

    def ddim_coef_table(self, device=None) -> torch.Tensor:
        """[timesteps, 4] fp32 rows {sqrt_abar, sqrt(1-abar), sqrt(abar_prev), sqrt(1-abar_prev)} with the
        last two formed in fp32 from the fp32-cast abar_prev, as the eta=0 update does (:1426-1437)."""
        sa, s1, abp = self.f32("sqrt_alphas_cumprod"), self.f32("sqrt_one_minus_alphas_cumprod"), self.f32("alphas_cumprod_prev")
        t = torch.stack([sa, s1, torch.sqrt(abp), torch.sqrt(1 - abp)], dim=1).contiguous()  # torch.sqrt, as upstream
        return t.to(device) if device is not None else t


class DDPMScheduler(ScheduleTables):
    """Cosine schedule with the getters of the reference class (latent_module.py:1241-1297).

    Getters return the fp32 value per sample shaped for broadcasting against `shape` (the reference
    materialises the full broadcast; the HIP kernels gather per sample instead)."""

    def __init__(self, timesteps: int, scale: float = 1.0):
        self.scale = scale
        super().__init__(get_named_beta_schedule("cosine", timesteps))

    def _get(self, name, t: torch.Tensor, shape):
        v = self.f32(name, t.device)[t.long()]
        return v.view(-1, *([1] * (len(shape) - 1)))

    def get_beta(self, t, shape):
        return self._get("betas", t, shape)

    def get_sqrt_alpha_cum(self, t, shape):
        return self._get("sqrt_alphas_cumprod", t, shape)

    def get_alpha_cum(self, t, shape):
        return self._get("alphas_cumprod", t, shape)

    def get_alpha_prev_cum(self, t, shape):
        return self._get("alphas_cumprod_prev", t, shape)

    def get_sqrt_one_minus_alpha_cum(self, t, shape):
        return self._get("sqrt_one_minus_alphas_cumprod", t, shape)

    def get_snr(self, t):
        sa = self.get_sqrt_alpha_cum(t, t.shape)
        s1 = self.get_sqrt_one_minus_alpha_cum(t, t.shape)
        return (sa ** 2) / (s1 ** 2)
