"""`GaussianDiffusion` with the reference's names and argument meaning (reference
fairseq/models/text_to_speech/diffusion/gaussian_diffusion.py:144-786), its per-element arithmetic in the HIP kernels
`dn_q_sample`, `dn_gaussian_step` and `dn_gaussian_moments`.  The float64 schedule tables live on the host exactly like upstream;
what the device sees is their fp32 cast (what `_extract_into_tensor` produces).  Scope: eps-prediction (the only mean type
`create_diffusion` builds besides START_X), FIXED_LARGE / FIXED_SMALL / LEARNED_RANGE variances, MSE and KL losses incl. the
learned-variance VB term.
`model` is any callable (x, t, **kw) -> tensor with channels on dim 1, returning fp32 CUDA tensors."""
import ctypes as C
import enum

import numpy as np
import torch

from .. import _lib
from ..scheduler import ScheduleTables, get_named_beta_schedule  # noqa: F401


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()
    START_X = enum.auto()
    EPSILON = enum.auto()


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()


class LossType(enum.Enum):
    MSE = enum.auto()
    RESCALED_MSE = enum.auto()
    KL = enum.auto()
    RESCALED_KL = enum.auto()

    def is_vb(self):
        return self in (LossType.KL, LossType.RESCALED_KL)


def mean_flat(t):
    return t.mean(dim=list(range(1, t.dim())))


class GaussianDiffusion(ScheduleTables):
    def __init__(self, *, betas, model_mean_type, model_var_type, loss_type):
        if model_mean_type != ModelMeanType.EPSILON:
            raise NotImplementedError("only eps-prediction is built (what the DiffNorm recipe and create_diffusion's default use)")
        if model_var_type == ModelVarType.LEARNED:
            raise NotImplementedError("ModelVarType.LEARNED is not reachable through create_diffusion")
        self.model_mean_type, self.model_var_type, self.loss_type = model_mean_type, model_var_type, loss_type
        super().__init__(betas)
        self._dev_tables = {}

    # ---- device tables
    def _fixed_log_variance(self):
        if self.model_var_type == ModelVarType.FIXED_LARGE:  # (:300-305)
            return np.log(np.append(self.posterior_variance[1], self.betas[1:]))
        return self.posterior_log_variance_clipped

    def _table(self, device):
        key = str(device)
        if key not in self._dev_tables:
            tab = self.gaussian_table(device, fixed_large=self.model_var_type == ModelVarType.FIXED_LARGE)  # (:300-310)
            sa, s1 = self.f32("sqrt_alphas_cumprod", device), self.f32("sqrt_one_minus_alphas_cumprod", device)
            self._dev_tables[key] = (tab, sa, s1)
        return self._dev_tables[key]

    @staticmethod
    def _t32(t, device):
        return t.to(device=device, dtype=torch.int32).contiguous()

    # ---- forward process
    def q_sample(self, x_start, t, noise=None):
        """q(x_t | x_0) (:215-230)."""
        lib = _lib.load()
        if noise is None:
            noise = torch.randn_like(x_start)
        assert noise.shape == x_start.shape
        x, nz = x_start.float().contiguous(), noise.float().contiguous()
        _, sa, s1 = self._table(x.device)
        out = torch.empty_like(x)
        N, inner = x.shape[0], x[0].numel()
        _lib.check(lib.dn_q_sample(x.data_ptr(), nz.data_ptr(), out.data_ptr(), None, _lib.DN_F32, inner, N, inner, inner, 1,
                                   sa.data_ptr(), s1.data_ptr(), self._t32(t, x.device).data_ptr(), _lib.current_stream()),
                   "dn_q_sample")
        return out

    # ---- reverse process
    def _step(self, model, x, t, noise, clip_denoised, sampler, eta=0.0, model_kwargs=None):
        lib = _lib.load()
        x = x.float().contiguous()
        out = model(x, t, **(model_kwargs or {}))
        if isinstance(out, tuple):
            out = out[0]
        out = out.float().contiguous()
        learned = self.model_var_type == ModelVarType.LEARNED_RANGE
        B, Cc = x.shape[:2]
        assert out.shape == ((B, Cc * 2, *x.shape[2:]) if learned else x.shape)
        tab, _, _ = self._table(x.device)
        sample, x0 = torch.empty_like(x), torch.empty_like(x)
        nz = noise.float().contiguous() if noise is not None else None
        p = _lib.GaussianStep(x.data_ptr(), out.data_ptr(), _lib.ptr(nz), sample.data_ptr(), x0.data_ptr(),
                              self._t32(t, x.device).data_ptr(), tab.data_ptr(), B, x[0].numel(), int(learned),
                              int(clip_denoised), sampler, float(eta))
        _lib.check(lib.dn_gaussian_step(C.byref(p), _lib.current_stream()), "dn_gaussian_step")
        return {"sample": sample, "pred_xstart": x0}

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, noise=None):
        """x_{t-1} ~ p(. | x_t) (:376-417).  `noise` may be injected; otherwise it is drawn like upstream (randn_like)."""
        if denoised_fn is not None or cond_fn is not None:
            raise NotImplementedError("denoised_fn / cond_fn hooks are not on the DiffNorm path")
        noise = torch.randn_like(x) if noise is None else noise
        return self._step(model, x, t, noise, clip_denoised, 0, model_kwargs=model_kwargs)

    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0, noise=None):
        """(:513-560)."""
        if denoised_fn is not None or cond_fn is not None:
            raise NotImplementedError("denoised_fn / cond_fn hooks are not on the DiffNorm path")
        noise = torch.randn_like(x) if noise is None else noise
        return self._step(model, x, t, noise, clip_denoised, 1, eta=eta, model_kwargs=model_kwargs)

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                  model_kwargs=None, device=None, progress=False):
        """(:459-511)."""
        if device is None:  # upstream takes the model's device (:484-485); a callable has none: follow the noise, else cuda
            device = noise.device if noise is not None else torch.device("cuda", torch.cuda.current_device())
        img = noise if noise is not None else torch.randn(*shape, device=device)
        for i in list(range(self.num_timesteps))[::-1]:
            t = torch.tensor([i] * shape[0], device=device)
            with torch.no_grad():
                out = self.p_sample(model, img, t, clip_denoised=clip_denoised, model_kwargs=model_kwargs)
                yield out
                img = out["sample"]

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                      device=None, progress=False):
        """(:419-457)."""
        final = None
        for sample in self.p_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised,
                                                     model_kwargs=model_kwargs, device=device):
            final = sample
        return final["sample"]

    # ---- moments (dn_gaussian_moments)
    def _moments(self, x, t, model_out=None, x_start=None, clip_denoised=True, want=()):
        lib = _lib.load()
        x = x.float().contiguous()
        tab, _, _ = self._table(x.device)
        outs = {k: torch.empty_like(x) for k in want}
        mo = model_out.float().contiguous() if model_out is not None else None
        xs = x_start.float().contiguous() if x_start is not None else None
        learned = self.model_var_type == ModelVarType.LEARNED_RANGE and mo is not None
        t32 = self._t32(t, x.device)
        g = lambda k: _lib.ptr(outs.get(k))
        p = _lib.GaussianMoments(x.data_ptr(), _lib.ptr(mo), _lib.ptr(xs), t32.data_ptr(), tab.data_ptr(), g("mean"), g("variance"),
                                 g("log_variance"), g("pred_xstart"), g("vb"), g("reverse_sample"), x.shape[0], x[0].numel(),
                                 int(learned), int(clip_denoised))
        _lib.check(lib.dn_gaussian_moments(C.byref(p), _lib.current_stream()), "dn_gaussian_moments")
        return outs

    def _model_out(self, model, x, t, model_kwargs):
        out = model(x, t, **(model_kwargs or {}))
        extra = None
        if isinstance(out, tuple):
            out, extra = out
        B, Cc = x.shape[:2]
        learned = self.model_var_type == ModelVarType.LEARNED_RANGE
        assert out.shape == ((B, Cc * 2, *x.shape[2:]) if learned else x.shape)
        return out, extra

    def q_posterior_mean_variance(self, x_start, x_t, t):
        """q(x_{t-1} | x_t, x_0) (:232-252) -> (mean, variance, log_variance_clipped), each shaped like x_t."""
        assert x_start.shape == x_t.shape
        o = self._moments(x_t, t, x_start=x_start, want=("mean", "variance", "log_variance"))
        return o["mean"], o["variance"], o["log_variance"]

    def p_mean_variance(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        """p(x_{t-1} | x_t) and the x_0 prediction (:254-332)."""
        if denoised_fn is not None:
            raise NotImplementedError("denoised_fn is not on the DiffNorm path")
        out, extra = self._model_out(model, x, t, model_kwargs)
        o = self._moments(x, t, model_out=out, clip_denoised=clip_denoised, want=("mean", "variance", "log_variance", "pred_xstart"))
        o["extra"] = extra
        return o

    def _predict_xstart_from_eps(self, x_t, t, eps):
        """(:334-339)."""
        keep = self.model_var_type
        self.model_var_type = ModelVarType.FIXED_SMALL  # eps carries no variance channels here
        try:
            return self._moments(x_t, t, model_out=eps, clip_denoised=False, want=("pred_xstart",))["pred_xstart"]
        finally:
            self.model_var_type = keep

    def ddim_reverse_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0):
        """x_{t+1} along the deterministic DDIM path (:562-598)."""
        assert eta == 0.0, "Reverse ODE only for deterministic path"
        if denoised_fn is not None or cond_fn is not None:
            raise NotImplementedError("denoised_fn / cond_fn hooks are not on the DiffNorm path")
        out, _ = self._model_out(model, x, t, model_kwargs)
        o = self._moments(x, t, model_out=out, clip_denoised=clip_denoised, want=("reverse_sample", "pred_xstart"))
        return {"sample": o["reverse_sample"], "pred_xstart": o["pred_xstart"]}

    def _vb_terms_bpd(self, model, x_start, x_t, t, clip_denoised=True, model_kwargs=None):
        """One term of the variational bound in bits per dimension (:682-713): per-sample means of the kernel's per-element
        KL / decoder-NLL values."""
        out, _ = self._model_out(model, x_t, t, model_kwargs)
        o = self._moments(x_t, t, model_out=out, x_start=x_start, clip_denoised=clip_denoised, want=("vb", "pred_xstart"))
        return {"output": mean_flat(o["vb"]), "pred_xstart": o["pred_xstart"]}

    def training_losses(self, model, x_start, t, model_kwargs=None, noise=None):
        """(:715-786) for eps-prediction: MSE / RESCALED_MSE (with the VB term of a learned variance, computed on the model's
        output as the reference computes it on the detached mean) and KL / RESCALED_KL.  Forward values (the HIP path of
        this scheduler has no autograd; the DiffNorm training loss is LatentDiscreteModel.forward)."""
        noise = torch.randn_like(x_start) if noise is None else noise
        x_t = self.q_sample(x_start, t, noise=noise)
        terms = {}
        if self.loss_type.is_vb():
            terms["loss"] = self._vb_terms_bpd(model, x_start, x_t, t, clip_denoised=False, model_kwargs=model_kwargs)["output"]
            if self.loss_type == LossType.RESCALED_KL:
                terms["loss"] = terms["loss"] * self.num_timesteps
            return terms
        out = model(x_t, t, **(model_kwargs or {}))
        misc = None
        if isinstance(out, tuple):
            out, misc = out
        terms["misc"] = misc
        out = out.float()
        if self.model_var_type == ModelVarType.LEARNED_RANGE:
            Cc = x_t.shape[1]
            assert out.shape == (x_t.shape[0], Cc * 2, *x_t.shape[2:])
            frozen = out.detach()
            terms["vb"] = self._vb_terms_bpd(lambda *a, **k: frozen, x_start, x_t, t, clip_denoised=False)["output"]
            if self.loss_type == LossType.RESCALED_MSE:
                terms["vb"] = terms["vb"] * (self.num_timesteps / 1000.0)
            out = out[:, :Cc]
        terms["mse"] = mean_flat((noise.to(out.device) - out) ** 2)
        terms["loss"] = terms["mse"] + terms["vb"] if "vb" in terms else terms["mse"]
        return terms
