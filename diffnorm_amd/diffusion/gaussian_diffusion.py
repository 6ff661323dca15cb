"""`GaussianDiffusion` with the reference's names and argument meaning (reference
fairseq/models/text_to_speech/diffusion/gaussian_diffusion.py:144-786), its per-element arithmetic in the HIP kernels
`dn_q_sample`, `dn_gaussian_step` and `dn_gaussian_moments`.  The float64 schedule tables live on the host exactly like upstream;
what the device sees is their fp32 cast (what `_extract_into_tensor` produces).  Scope: what `create_diffusion` can build --
EPSILON and START_X mean types, FIXED_LARGE / FIXED_SMALL / LEARNED_RANGE variances, MSE and KL losses incl. the learned-variance
VB term -- with the samplers (p_sample / ddim_sample and their loops, ddim_reverse_sample), `cond_fn` guidance (condition_mean /
condition_score) and the bits-per-dim evaluation (_prior_bpd, calc_bpd_loop).
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
        if model_mean_type not in (ModelMeanType.EPSILON, ModelMeanType.START_X):
            raise NotImplementedError("ModelMeanType.PREVIOUS_X is not reachable through create_diffusion")
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
    @property
    def _start_x(self) -> int:
        return int(self.model_mean_type == ModelMeanType.START_X)

    def _through_denoised_fn(self, out, x, t, denoised_fn):
        """`denoised_fn`: "a function which applies to the x_start prediction before it is used to sample; applies before
        clip_denoised" (:263-265, process_xstart :309-314) -- a host callable between the two halves of the fused step.  Un-fused
        fallback: the kernel first produces the UNCLIPPED x_0 prediction of the model output (whatever its mean type), the
        callable maps it, and the result goes back in as a START_X model output (with the learned-variance channels it came
        with), so the second kernel call clips it, forms the posterior mean and samples exactly as upstream does from the
        processed prediction.  -> (model output to hand to the kernel, start_x flag)."""
        x0 = self._moments(x, t, model_out=out, clip_denoised=False, want=("pred_xstart",))["pred_xstart"]
        x0 = denoised_fn(x0).float().contiguous()
        assert x0.shape == x.shape, "denoised_fn must keep the shape of the x_start prediction"
        if self.model_var_type == ModelVarType.LEARNED_RANGE:
            return torch.cat([x0, out[:, x.shape[1]:]], dim=1).contiguous(), 1
        return x0, 1

    def _step(self, model, x, t, noise, clip_denoised, sampler, eta=0.0, model_kwargs=None, cond_fn=None, denoised_fn=None):
        lib = _lib.load()
        x = x.float().contiguous()
        out = model(x, t, **(model_kwargs or {}))
        if isinstance(out, tuple):
            out = out[0]
        out = out.float().contiguous()
        learned = self.model_var_type == ModelVarType.LEARNED_RANGE
        B, Cc = x.shape[:2]
        assert out.shape == ((B, Cc * 2, *x.shape[2:]) if learned else x.shape)
        start_x = self._start_x
        if denoised_fn is not None:
            out, start_x = self._through_denoised_fn(out, x, t, denoised_fn)
        tab, _, _ = self._table(x.device)
        sample, x0 = torch.empty_like(x), torch.empty_like(x)
        nz = noise.float().contiguous() if noise is not None else None
        # cond_fn(x, t, **model_kwargs) = grad log p(y | x): it depends on (x, t) only, so it is evaluated here and the kernel applies
        # condition_mean (p_sample, :346-358) or condition_score (ddim_sample, :360-374) per element
        grad = cond_fn(x, t, **(model_kwargs or {})).float().contiguous() if cond_fn is not None else None
        assert grad is None or grad.shape == x.shape
        p = _lib.GaussianStep(x.data_ptr(), out.data_ptr(), _lib.ptr(nz), sample.data_ptr(), x0.data_ptr(),
                              self._t32(t, x.device).data_ptr(), tab.data_ptr(), B, x[0].numel(), int(learned),
                              int(clip_denoised), sampler, float(eta), start_x, _lib.ptr(grad))
        _lib.check(lib.dn_gaussian_step(C.byref(p), _lib.current_stream()), "dn_gaussian_step")
        return {"sample": sample, "pred_xstart": x0}

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, noise=None):
        """x_{t-1} ~ p(. | x_t) (:376-417).  `noise` may be injected; otherwise it is drawn like upstream (randn_like)."""
        noise = torch.randn_like(x) if noise is None else noise
        return self._step(model, x, t, noise, clip_denoised, 0, model_kwargs=model_kwargs, cond_fn=cond_fn, denoised_fn=denoised_fn)

    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0, noise=None):
        """(:513-560)."""
        noise = torch.randn_like(x) if noise is None else noise
        return self._step(model, x, t, noise, clip_denoised, 1, eta=eta, model_kwargs=model_kwargs, cond_fn=cond_fn, denoised_fn=denoised_fn)

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                  model_kwargs=None, device=None, progress=False):
        """(:459-511)."""
        if device is None:  # upstream takes the model's device (:484-485); a callable has none: follow the noise, else cuda
            device = noise.device if noise is not None else torch.device("cuda", torch.cuda.current_device())
        img = noise if noise is not None else torch.randn(*shape, device=device)
        for i in list(range(self.num_timesteps))[::-1]:
            t = torch.tensor([i] * shape[0], device=device)
            with torch.no_grad():
                out = self.p_sample(model, img, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn, cond_fn=cond_fn, model_kwargs=model_kwargs)
                yield out
                img = out["sample"]

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                      device=None, progress=False):
        """(:419-457)."""
        final = None
        for sample in self.p_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                                     cond_fn=cond_fn, model_kwargs=model_kwargs, device=device):
            final = sample
        return final["sample"]

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                     model_kwargs=None, device=None, progress=False, eta=0.0):
        """(:631-680)."""
        if device is None:
            device = noise.device if noise is not None else torch.device("cuda", torch.cuda.current_device())
        img = noise if noise is not None else torch.randn(*shape, device=device)
        for i in list(range(self.num_timesteps))[::-1]:
            t = torch.tensor([i] * shape[0], device=device)
            with torch.no_grad():
                out = self.ddim_sample(model, img, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn, cond_fn=cond_fn,
                                       model_kwargs=model_kwargs, eta=eta)
                yield out
                img = out["sample"]

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                         device=None, progress=False, eta=0.0):
        """(:600-629)."""
        final = None
        for sample in self.ddim_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                                        cond_fn=cond_fn, model_kwargs=model_kwargs, device=device, eta=eta):
            final = sample
        return final["sample"]

    # ---- moments (dn_gaussian_moments)
    def _moments(self, x, t, model_out=None, x_start=None, clip_denoised=True, want=(), learned=None, start_x=None):
        lib = _lib.load()
        x = x.float().contiguous()
        tab, _, _ = self._table(x.device)
        outs = {k: torch.empty_like(x) for k in want}
        mo = model_out.float().contiguous() if model_out is not None else None
        xs = x_start.float().contiguous() if x_start is not None else None
        learned = (self.model_var_type == ModelVarType.LEARNED_RANGE if learned is None else learned) and mo is not None
        t32 = self._t32(t, x.device)
        g = lambda k: _lib.ptr(outs.get(k))
        p = _lib.GaussianMoments(x.data_ptr(), _lib.ptr(mo), _lib.ptr(xs), t32.data_ptr(), tab.data_ptr(), g("mean"), g("variance"),
                                 g("log_variance"), g("pred_xstart"), g("vb"), g("reverse_sample"), x.shape[0], x[0].numel(),
                                 int(learned), int(clip_denoised), int(self._start_x if start_x is None else start_x))
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
        out, extra = self._model_out(model, x, t, model_kwargs)
        start_x = None
        if denoised_fn is not None:
            out, start_x = self._through_denoised_fn(out, x.float().contiguous(), t, denoised_fn)
            start_x = bool(start_x)
        o = self._moments(x, t, model_out=out, clip_denoised=clip_denoised, want=("mean", "variance", "log_variance", "pred_xstart"), start_x=start_x)
        o["extra"] = extra
        return o

    def _predict_xstart_from_eps(self, x_t, t, eps):
        """(:334-339)."""
        # eps carries no variance channels here and IS an eps whatever the model's mean type (explicit arguments: no state is mutated)
        return self._moments(x_t, t, model_out=eps, clip_denoised=False, want=("pred_xstart",), learned=False, start_x=False)["pred_xstart"]

    def _predict_eps_from_xstart(self, x_t, t, pred_xstart):
        """(:341-344)."""
        tab, _, _ = self._table(x_t.device)
        rows = tab[self._t32(t, x_t.device).long()]
        shape = (-1,) + (1,) * (x_t.dim() - 1)
        return (rows[:, 0].view(shape) * x_t.float() - pred_xstart.float()) / rows[:, 1].view(shape)

    def q_mean_variance(self, x_start, t):
        """q(x_t | x_0) (:203-213) -> (mean, variance, log_variance), broadcast like upstream's _extract_into_tensor."""
        dev = x_start.device
        idx = t.to(dev).long()
        shape = (-1,) + (1,) * (x_start.dim() - 1)
        ones = torch.ones_like(x_start, dtype=torch.float32)
        mean = self.f32("sqrt_alphas_cumprod", dev)[idx].view(shape) * x_start.float()
        var = self.f32(1.0 - self.alphas_cumprod, dev)[idx].view(shape) * ones
        logv = self.f32("log_one_minus_alphas_cumprod", dev)[idx].view(shape) * ones
        return mean, var, logv

    def condition_mean(self, cond_fn, p_mean_var, x, t, model_kwargs=None):
        """(:346-358)."""
        gradient = cond_fn(x, t, **(model_kwargs or {}))
        return p_mean_var["mean"].float() + p_mean_var["variance"] * gradient.float()

    def condition_score(self, cond_fn, p_mean_var, x, t, model_kwargs=None):
        """(:360-374)."""
        dev = x.device
        shape = (-1,) + (1,) * (x.dim() - 1)
        alpha_bar = self.f32("alphas_cumprod", dev)[t.to(dev).long()].view(shape)
        eps = self._predict_eps_from_xstart(x, t, p_mean_var["pred_xstart"])
        eps = eps - (1 - alpha_bar).sqrt() * cond_fn(x, t, **(model_kwargs or {}))
        out = dict(p_mean_var)
        out["pred_xstart"] = self._predict_xstart_from_eps(x, t, eps)
        out["mean"], _, _ = self.q_posterior_mean_variance(x_start=out["pred_xstart"], x_t=x, t=t)
        return out

    def _prior_bpd(self, x_start):
        """The prior KL term of the variational bound in bits per dimension (:788-806): KL(q(x_T | x_0) || N(0, I))."""
        B = x_start.shape[0]
        t = torch.tensor([self.num_timesteps - 1] * B, device=x_start.device)
        mean, _, logv = self.q_mean_variance(x_start, t)
        kl = 0.5 * (-1.0 - logv + torch.exp(logv) + mean ** 2)  # normal_kl(mean, logv, 0, 0) (diffusion_utils.py:10-34)
        return mean_flat(kl) / np.log(2.0)

    def calc_bpd_loop(self, model, x_start, clip_denoised=True, model_kwargs=None, noises=None):
        """The whole variational bound, one term per timestep (:808-858).  `noises[t]` may be injected (parity runs); otherwise
        drawn like upstream."""
        dev = x_start.device
        B = x_start.shape[0]
        vb, xstart_mse, mse = [], [], []
        for t in list(range(self.num_timesteps))[::-1]:
            t_batch = torch.tensor([t] * B, device=dev)
            noise = torch.randn_like(x_start) if noises is None else noises[t].to(dev)
            x_t = self.q_sample(x_start=x_start, t=t_batch, noise=noise)
            with torch.no_grad():
                out = self._vb_terms_bpd(model, x_start=x_start, x_t=x_t, t=t_batch, clip_denoised=clip_denoised, model_kwargs=model_kwargs)
            vb.append(out["output"])
            xstart_mse.append(mean_flat((out["pred_xstart"] - x_start) ** 2))
            eps = self._predict_eps_from_xstart(x_t, t_batch, out["pred_xstart"])
            mse.append(mean_flat((eps - noise) ** 2))
        vb, xstart_mse, mse = torch.stack(vb, dim=1), torch.stack(xstart_mse, dim=1), torch.stack(mse, dim=1)
        prior_bpd = self._prior_bpd(x_start)
        return {"total_bpd": vb.sum(dim=1) + prior_bpd, "prior_bpd": prior_bpd, "vb": vb, "xstart_mse": xstart_mse, "mse": mse}

    def ddim_reverse_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0):
        """x_{t+1} along the deterministic DDIM path (:562-598)."""
        assert eta == 0.0, "Reverse ODE only for deterministic path"
        out, _ = self._model_out(model, x, t, model_kwargs)
        if denoised_fn is not None and cond_fn is None:
            out, _ = self._through_denoised_fn(out, x.float().contiguous(), t, denoised_fn)
            o = self._moments(x, t, model_out=out, clip_denoised=clip_denoised, want=("reverse_sample", "pred_xstart"), start_x=True)
            return {"sample": o["reverse_sample"], "pred_xstart": o["pred_xstart"]}
        if denoised_fn is not None:
            out, sx = self._through_denoised_fn(out, x.float().contiguous(), t, denoised_fn)
            pmv = self._moments(x, t, model_out=out, clip_denoised=clip_denoised, want=("mean", "variance", "log_variance", "pred_xstart"), start_x=True)
            x0 = self.condition_score(cond_fn, pmv, x, t, model_kwargs=model_kwargs)["pred_xstart"]
            o = self._moments(x, t, model_out=x0, clip_denoised=False, want=("reverse_sample", "pred_xstart"), learned=False, start_x=True)
            return {"sample": o["reverse_sample"], "pred_xstart": o["pred_xstart"]}
        if cond_fn is not None:  # (:579-580): the conditioned x_0 prediction, then the reverse-ODE step from it (START_X form)
            pmv = self._moments(x, t, model_out=out, clip_denoised=clip_denoised, want=("mean", "variance", "log_variance", "pred_xstart"))
            x0 = self.condition_score(cond_fn, pmv, x, t, model_kwargs=model_kwargs)["pred_xstart"]
            o = self._moments(x, t, model_out=x0, clip_denoised=False, want=("reverse_sample", "pred_xstart"), learned=False, start_x=True)
            return {"sample": o["reverse_sample"], "pred_xstart": o["pred_xstart"]}
        o = self._moments(x, t, model_out=out, clip_denoised=clip_denoised, want=("reverse_sample", "pred_xstart"))
        return {"sample": o["reverse_sample"], "pred_xstart": o["pred_xstart"]}

    def _vb_terms_bpd(self, model, x_start, x_t, t, clip_denoised=True, model_kwargs=None):
        """One term of the variational bound in bits per dimension (:682-713): per-sample means of the kernel's per-element
        KL / decoder-NLL values."""
        out, _ = self._model_out(model, x_t, t, model_kwargs)
        o = self._moments(x_t, t, model_out=out, x_start=x_start, clip_denoised=clip_denoised, want=("vb", "pred_xstart"))
        return {"output": mean_flat(o["vb"]), "pred_xstart": o["pred_xstart"]}

    def training_losses(self, model, x_start, t, model_kwargs=None, noise=None):
        """(:715-786): MSE / RESCALED_MSE (with the VB term of a learned variance, computed on the model's
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
        target = x_start.float() if self._start_x else noise  # (:768-776)
        terms["mse"] = mean_flat((target.to(out.device) - out) ** 2)
        terms["loss"] = terms["mse"] + terms["vb"] if "vb" in terms else terms["mse"]
        return terms
