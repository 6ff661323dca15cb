"""Host-side mirror of the reference's hot-path modules (same names, constructor arguments, method
signatures, return values and state-dict keys as fairseq/models/text_to_speech/latent_module.py upstream),
computing through libdiffnorm_hip.so.

The classes are `torch.nn.Module`s only so that they own `Parameter`s under the reference's state-dict
keys and plug into fairseq unchanged; every `forward` dispatches to the C-ABI engine.
There is no CPU path: calling them without a HIP device raises `DiffNormHipError`.

Random draws follow the reference: the VAE posterior noise comes from torch's *CPU* generator with
shape [B, z, T] (reference distributions.py:38-40), the DDIM start noise from the device generator
(latent_module.py:1409); both can be injected for parity runs.
"""
from typing import Dict, List, Optional

import torch
from torch import nn

from . import _lib, engine, ops, scheduler, synthetic
from .scheduler import DDPMScheduler  # re-exported under its reference name  # noqa: F401


def exists(x):
    return x is not None


def lengths_to_mask(lengths: torch.Tensor, max_len: Optional[int] = None) -> torch.Tensor:
    """`arange < len` (reference fairseq/data/data_utils.py:542-552)."""
    max_len = int(lengths.max()) if max_len is None else max_len
    return torch.arange(max_len, device=lengths.device).view(1, -1) < lengths.view(-1, 1)


def _mask_to_lengths(mask: torch.Tensor) -> torch.Tensor:
    """The path only ever sees right-padded masks (lengths_to_mask); the kernels take lengths."""
    lengths = mask.sum(dim=1)
    if not torch.equal(mask, lengths_to_mask(lengths, mask.shape[1])):
        raise ValueError("input_mask must be a right-padded mask (arange < length)")
    return lengths


def label_smoothed_nll_loss(lprobs, target, epsilon, ignore_index=None, reduce=True):
    """reference fairseq/criterions/label_smoothed_cross_entropy.py:34-51 (device-side reductions)."""
    if target.dim() == lprobs.dim() - 1:
        target = target.unsqueeze(-1)
    nll = -lprobs.gather(dim=-1, index=target)
    smooth = -lprobs.sum(dim=-1, keepdim=True)
    if ignore_index is not None:
        pad = target.eq(ignore_index)
        nll = nll.masked_fill(pad, 0.0)
        smooth = smooth.masked_fill(pad, 0.0)
    else:
        nll, smooth = nll.squeeze(-1), smooth.squeeze(-1)
    if reduce:
        nll, smooth = nll.sum(), smooth.sum()
    eps_i = epsilon / (lprobs.size(-1) - 1)
    return (1.0 - epsilon - eps_i) * nll + eps_i * smooth, nll


class _ParamTree(nn.Module):
    """Owns parameters under dotted state-dict keys by growing anonymous sub-modules on demand."""

    def _attach(self, key: str, tensor: torch.Tensor, buffer: bool = False):
        node = self
        parts = key.split(".")
        for name in parts[:-1]:
            if name not in node._modules:
                node.add_module(name, nn.Module())
            node = node._modules[name]
        if buffer:
            node.register_buffer(parts[-1], tensor)
        else:
            node.register_parameter(parts[-1], nn.Parameter(tensor))

    def _adopt(self, tensors: Dict[str, torch.Tensor]):
        for k, v in tensors.items():
            self._attach(k, v)
        self._engine = None
        self._engine_key = None

    def _state_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    @property
    def device(self):
        return next(self.parameters()).device


class Model(_ParamTree):
    """eps-predictor (reference latent_module.py:709-876): WaveNet (FiLM time conditioning) -> sinusoidal
    positions -> time-conditioned transformer -> Linear.  Same constructor keywords as upstream."""

    def __init__(self, dim, latent_dim, *, depth=12, dim_head=64, heads=8, ff_mult=4, wavenet_layers=8, wavenet_stacks=4,
                 dim_cond_mult=4, use_flash_attn=False, dim_prompt=None, num_latents_m=64, resampler_depth=2,
                 cond_drop_prob=0., condition_on_prompt=False, dtype="bf16", seed=0):
        super().__init__()
        if ff_mult != 4:
            raise NotImplementedError("the engine packs ff_mult = 4 (the only value the recipe uses)")
        if condition_on_prompt and not dim_prompt:
            raise ValueError("condition_on_prompt needs dim_prompt")
        self.dim, self.latent_dim = dim, latent_dim
        self.cond_drop_prob = cond_drop_prob
        self.condition_on_prompt = bool(condition_on_prompt)
        self.cfg = synthetic.eps_config(dim, latent_dim, depth, heads, dim_head, wavenet_layers, wavenet_stacks, dim_cond_mult,
                                        dim_prompt=dim_prompt if condition_on_prompt else 0, num_latents_m=num_latents_m,
                                        resampler_depth=resampler_depth)
        self.arith = dtype
        self._adopt(synthetic.random_eps_state_dict(self.cfg, seed))
        self._attach("pos_embed._float_tensor", torch.zeros(1), buffer=True)  # key present upstream (:774-779)
        if self.condition_on_prompt:
            self._attach("perceiver_resampler.embed_positions._float_tensor", torch.zeros(1), buffer=True)  # (:428-435)

    # ---- training: set by LatentDiscreteModel.enable_training -- the flat master buffer of the diffusion
    # training engine becomes this module's only parameter; state_dict() keeps the reference's keys
    _train_engine = None

    def _adopt_flat(self, eng):
        for name in list(self._modules):
            del self._modules[name]
        self._buffers.clear()
        self.flat_params = nn.Parameter(eng.master)
        self.flat_params.grad = eng.grads
        self._train_engine = eng
        self._engine = None

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        if self._train_engine is None:
            return super().state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        from collections import OrderedDict

        out = OrderedDict() if destination is None else destination
        for k, v in self._train_engine.state_dict().items():
            out[prefix + k] = v
        out[prefix + "pos_embed._float_tensor"] = torch.zeros(1)
        return out

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        if self._train_engine is None:
            return super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
        want = set(self._train_engine.state_dict())
        got = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix) and not k.endswith("pos_embed._float_tensor")}
        missing_keys += [prefix + k for k in want - set(got)]
        unexpected_keys += [prefix + k for k in set(got) - want]
        if not (want - set(got)):
            self._train_engine.load_state_dict({k: got[k] for k in want})

    def engine(self) -> engine.EpsEngine:
        if self._train_engine is not None:
            key = ("train", self._train_engine.update_count)
            if self._engine is None or self._engine_key != key:
                self._engine = engine.EpsEngine(self._train_engine.state_dict(), self.cfg, dtype=self.arith, device=self.device)
                self._engine_key = key
            return self._engine
        key = self._state_key()
        if self._engine is None or self._engine_key != key:
            sd = {k: v.detach().cpu() for k, v in self.state_dict().items() if not k.endswith("._float_tensor")}
            self._engine = engine.EpsEngine(sd, self.cfg, dtype=self.arith, device=self.device)
            self._engine_key = key
        return self._engine

    def forward(self, x, times, prompt=None, prompt_mask=None, input_mask=None, cond=None, cond_drop_prob=None, drop_mask=None):
        """x [B,T,latent], times [B] (raw integer steps), input_mask [B,T] bool -> eps_hat [B,T,latent].  With
        condition_on_prompt: prompt [B,Tp,dim_prompt], prompt_mask [B,Tp]; the classifier-free-guidance drop mask is drawn per
        sample with probability cond_drop_prob like upstream's prob_mask_like (:843) unless `drop_mask` [B] bool is given."""
        if input_mask is None:
            input_mask = torch.ones(x.shape[:2], dtype=torch.bool, device=x.device)
        lengths = _mask_to_lengths(input_mask)
        if self.condition_on_prompt:
            if prompt is None or prompt_mask is None:
                raise ValueError("this Model is conditioned on a prompt: pass prompt and prompt_mask")
            B = x.shape[0]
            if drop_mask is None:
                p = self.cond_drop_prob if cond_drop_prob is None else cond_drop_prob
                drop_mask = torch.ones(B, dtype=torch.bool) if p >= 1 else (
                    torch.zeros(B, dtype=torch.bool) if p <= 0 else torch.zeros(B).float().uniform_(0, 1) < p)
            return self.engine().forward_cond(x, times, lengths, prompt, _mask_to_lengths(prompt_mask), drop_mask)
        shared = bool((times == times[0]).all())
        return self.engine().forward(x, times, lengths, shared_t=shared)

    def forward_with_cond_scale(self, *args, cond_scale=1., **kwargs):
        """null + (cond - null) * cond_scale (:813-826); without a prompt branch guidance is the identity."""
        if not self.condition_on_prompt:
            return self.forward(*args, **kwargs)
        kwargs.pop("cond_drop_prob", None)
        logits = self.forward(*args, cond_drop_prob=0., **kwargs)
        if cond_scale == 1.:
            return logits
        null = self.forward(*args, cond_drop_prob=1., **kwargs)
        return torch.add(null, logits - null, alpha=cond_scale)


def _set_dropout(engine, p: float):
    """Train mode of the reference's Attention(dropout=0.1) (latent_module.py:338,668): the mask's seed comes off torch's CPU
    generator, so `torch.manual_seed` (fairseq: seed + num_updates before every update) makes a step repeatable."""
    engine.attn_dropout = float(p)
    if p > 0.0:
        engine.dropout_seed, engine._dropout_calls = int(torch.randint(0, 2 ** 31 - 1, (1,))), 0


def _prepare_step(flat, eng, holder) -> bool:
    """Before a training forward.  (1) The master buffer may have been updated by an optimizer that is not the HIP one: the engine's
    bf16 working copy and its transposed weights are then refreshed from it.  Nothing observable tells that such an update
    happened -- fairseq's Adam writes through `p.data` (fairseq/optim/adam.py:185-236: `p_data_fp32 = p.data; ...addcdiv_`), which
    does NOT move the parameter's version counter (round 3 relied on it and trained nothing under `--optimizer adam`) -- so the
    engine keeps a flag instead: `work_current` is set by the two calls that bring the copies up to date (`refresh` after the
    HIP-backed FlatOptimizer's step, `sync_work`) and cleared by every backward of this bridge, after which any optimizer may
    move the master buffer.  Under FlatOptimizer the flag is set again before the next forward and nothing extra runs; under an
    external optimizer every forward that follows a backward pays one sync (fp32 -> bf16 + the transposes: what an update
    needs anyway).  (2) fairseq's FairseqOptimizer.zero_grad sets `p.grad = None`
    (fairseq/optim/fairseq_optimizer.py:129-133), which drops the alias `flat_params.grad is engine.grads`: that IS the zeroing,
    so the engine's gradient buffer is cleared here and the alias is restored by the backward.  -> whether the alias was dropped."""
    if not getattr(eng, "work_current", False):
        eng.sync_work()
    dropped = flat.grad is None
    if dropped:
        eng.zero_grad()
    return dropped


def _scaled_backward(eng, c: float, fresh: bool, run):
    """The engine's backward ADDS d loss / d theta into its gradient buffer; an upstream gradient c != 1 (fairseq's fp16 / amp loss
    scale, `optimizer.backward(loss)` on a scaled loss) must add c times that: g_old + c g = c (g_old / c + g) -- two elementwise
    passes over the flat buffer around the same kernels, exact for the power-of-two scales loss scalers use; a buffer that was
    just zeroed (`fresh`) skips the first pass."""
    if c == 1.0:
        return run()
    if not fresh:
        eng.grads.div_(c)
    run()
    eng.grads.mul_(c)


def _finish_backward(flat, eng, holder, run, c: float, fresh: bool):
    """What the bridge's backward hands back for `flat_params`, after `run` has ADDED d loss / d theta into the engine's buffer.

    Default: nothing -- `flat_params.grad` IS the engine's gradient buffer (the alias is restored when a zero_grad had dropped
    it), so any elementwise optimizer and an explicit flat all-reduce (fairseq's legacy_ddp) work on it in place.

    `holder._grads_through_autograd` (set by the plugin's train_step when the model arrives wrapped in
    torch.nn.parallel.DistributedDataParallel, fairseq's DEFAULT `--ddp-backend pytorch_ddp`: fairseq/dataclass/configs.py:301-309,
    fairseq/models/distributed_fairseq_model.py:59-84): torch-DDP's reducer hangs on the parameter's gradient ACCUMULATOR, which
    only fires when a gradient for the parameter comes out of the autograd graph.  So this micro-batch's gradient is computed into
    the (zeroed) engine buffer and returned as the Function's gradient for `flat_params`: autograd accumulates it into
    `flat_params.grad` (one copy of the buffer), the reducer's hook fires and all-reduces it like any other parameter's.  A
    zero upstream gradient (fairseq's ignore_grad dummy batches) returns zeros rather than nothing: every rank must feed the
    reducer in every step."""
    if not getattr(holder, "_grads_through_autograd", False):
        if c != 0.0:
            _scaled_backward(eng, c, fresh, run)
        if flat.grad is None:
            flat.grad = eng.grads
        eng.work_current = False  # an optimizer step may follow: see _prepare_step
        return None
    if flat.grad is eng.grads:  # an alias left from steps taken before the wrapper appeared: detach the accumulated gradient from it
        flat.grad = eng.grads.clone()
    eng.grads.zero_()
    if c != 0.0:
        run()
    out = eng.grads.clone()
    if c not in (0.0, 1.0):
        out.mul_(c)
    eng.work_current = False
    return out


def wrapped_by_torch_ddp(model) -> bool:
    """Is `model` (or something it wraps: fairseq's ModuleProxyWrapper keeps the DDP module under `.module`) a torch
    DistributedDataParallel?"""
    m, seen = model, 0
    while m is not None and seen < 8:
        if isinstance(m, torch.nn.parallel.DistributedDataParallel):
            return True
        m, seen = getattr(m, "module", None), seen + 1
    return False


class _VaeStepFn(torch.autograd.Function):
    """Autograd node around the HIP training engine: forward = dn_vae_train_forward (activations stay in the engine's
    workspace), backward = dn_vae_train_backward, which ADDS the parameter gradients into the flat gradient buffer that is
    `flat_params.grad` (so nothing is returned for the parameter; the alias is restored when a zero_grad had set it to None).
    Outputs: stats [8] (loss, nll, mse, kl, acc, ...) and the logits.  Two ways to differentiate it: through stats[0], the
    criterion's own loss (fused LS-CE gradient; any scalar upstream gradient: 1, a loss scale, or 0 for fairseq's ignore_grad), or
    through (stats[2], logits, stats[3]) = (mse_loss, lm_logits, kl_loss), the reference model's return values, for a caller that
    builds its loss itself."""

    @staticmethod
    def forward(ctx, flat, module, feat, units, lengths, noise, ntokens):
        eng = module._train_engine
        ctx.fresh = _prepare_step(flat, eng, module)
        stats, logits, _ = eng.forward(feat, units, lengths, noise=noise, ntokens=ntokens, want_logits=True)
        ctx.module, ctx.flat = module, flat
        return stats.clone(), logits

    @staticmethod
    def backward(ctx, g_stats, g_logits):
        eng = ctx.module._train_engine
        gs = [0.0] * 8 if g_stats is None else [float(v) for v in g_stats.tolist()]  # one host read of 8 floats
        if gs[0] != 0.0:
            if any(v != 0.0 for v in gs[1:]):
                raise NotImplementedError("differentiate the HIP VAE either through its fused criterion loss (stats[0]) or through (mse, logits, kl), not both")
            g_flat = _finish_backward(ctx.flat, eng, ctx.module, eng.backward, gs[0], ctx.fresh)
        elif gs[2] != 0.0 or gs[3] != 0.0 or (g_logits is not None and bool(g_logits.ne(0).any())):  # zero: fairseq's ignore_grad
            ext = g_logits if g_logits is not None else torch.zeros_like(eng._keep[5])
            g_flat = _finish_backward(ctx.flat, eng, ctx.module, lambda: eng.backward(ext_dlogits=ext, d_mse=gs[2], d_kl=gs[3]), 1.0, True)
        else:
            g_flat = _finish_backward(ctx.flat, eng, ctx.module, eng.backward, 0.0, ctx.fresh)
        return g_flat, None, None, None, None, None, None


class _EpsStepFn(torch.autograd.Function):
    """Autograd node around the HIP diffusion training engine (dn_eps_train_forward / _backward): the loss dict's total_loss is
    stats[0]; its backward ADDS the eps-predictor's gradients into `flat_params.grad` (nothing is returned for the parameter) for
    any scalar upstream gradient (1, a loss scale, 0 = ignore_grad)."""

    @staticmethod
    def forward(ctx, flat, owner, feat, units, lengths, z, times, jitter, true_noise):
        ctx.fresh = _prepare_step(flat, owner._train_engine, owner)
        stats = owner._train_engine.forward(feat, units, lengths, z, times, jitter, true_noise)
        ctx.owner, ctx.flat = owner, flat
        return stats.clone()

    @staticmethod
    def backward(ctx, g_stats):
        gs = [float(v) for v in g_stats.tolist()]
        if any(v != 0.0 for v in gs[1:]):
            raise NotImplementedError("the HIP diffusion loss is differentiated through total_loss")
        eng = ctx.owner._train_engine
        return (_finish_backward(ctx.flat, eng, ctx.owner, eng.backward, gs[0], ctx.fresh),) + (None,) * 8


class SpeechVAEEncoderDecoder(_ParamTree):
    """reference latent_module.py:1035-1142 (WaveNet encoder -> diagonal Gaussian -> WaveNet + transformer decoder
    -> 1004-way unit logits).  `latent_dim` is the upstream constructor flag (16 / 32 / 128)."""

    def __init__(self, dim=768, latent_dim=16, dtype="bf16", seed=1):
        super().__init__()
        self.dim, self.latent_dim = dim, latent_dim
        self.arith = dtype
        self._train_engine = None
        self.train_on_move = False  # the plugin's build_model sets it for a training run: see _apply
        self.attn_dropout = 0.1  # Attention(dropout=0.1) of the decoder transformer (:668); active in train() mode with the training engine
        self._adopt(synthetic.random_vae_state_dict(dim, latent_dim, seed=seed))

    def max_positions(self):
        return None

    def latent_channels(self) -> int:
        """Width of the posterior sample: dim / prod(chan_mults) / 2 (:1044-1051, 1067-1070)."""
        from .packing import vae_mults

        z = self.dim
        for m in vae_mults(self.latent_dim):
            z //= m
        return z // 2

    def _apply(self, fn, recurse=True):
        """`model.to(device)`: in a training run (train_on_move) the switch to the training engine happens HERE, because the
        reference trainer moves the model first and then builds its optimizer from `model.parameters()` (fairseq/trainer.py:
        292) -- long before the first train_step: the optimizer must see `flat_params`, not per-tensor parameters that a later
        switch would delete."""
        out = super()._apply(fn, recurse)
        if self.train_on_move and self._train_engine is None and self.device.type == "cuda":
            self.enable_training()
        return out

    # ---- training: the flat packed master buffer of the HIP training engine becomes THE parameter ----
    def enable_training(self):
        """Switches the module to the training engine (diffnorm_amd/training.py): its per-tensor parameters are replaced by
        one flat `flat_params` Parameter that aliases the engine's fp32 master buffer, with `.grad` aliasing the engine's
        gradient buffer -- any elementwise optimizer (fairseq's Adam included), gradient clipping by global norm and an
        explicit flat all-reduce (fairseq's legacy_ddp) work on it unchanged.  `state_dict()` / `load_state_dict()` keep
        speaking the reference's key layout."""
        if self._train_engine is not None:
            return self._train_engine
        from . import training

        sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
        dev = self.device
        eng = training.VaeTrainEngine(sd, dim=self.dim, latent_dim=self.latent_dim, dtype=self.arith, device=dev)
        for name in list(self._modules):
            del self._modules[name]
        self.flat_params = nn.Parameter(eng.master)
        self.flat_params.grad = eng.grads
        self._train_engine = eng
        self._engine = None
        return eng

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        if self._train_engine is None:
            return super().state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        from collections import OrderedDict

        out = OrderedDict() if destination is None else destination
        for k, v in self._train_engine.state_dict().items():
            out[prefix + k] = v
        return out

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        if self._train_engine is None:
            return super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
        want = set(self._train_engine.state_dict())
        got = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
        missing_keys += [prefix + k for k in want - set(got)]
        unexpected_keys += [prefix + k for k in set(got) - want]
        if not (want - set(got)):
            self._train_engine.load_state_dict({k: got[k] for k in want})

    def engine(self) -> engine.VaeEngine:
        if self._train_engine is not None:  # inference engine rebuilt from the master buffer when an update has happened
            key = ("train", self._train_engine.update_count)
            if self._engine is None or self._engine_key != key:
                self._engine = engine.VaeEngine(self._train_engine.state_dict(), dim=self.dim, latent_dim=self.latent_dim,
                                                dtype=self.arith, device=self.device)
                self._engine_key = key
            return self._engine
        key = self._state_key()
        if self._engine is None or self._engine_key != key:
            sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
            self._engine = engine.VaeEngine(sd, dim=self.dim, latent_dim=self.latent_dim, dtype=self.arith, device=self.device)
            self._engine_key = key
        return self._engine

    def _posterior_noise(self, B, T, noise):
        e = self.engine()
        if noise is None:  # CPU generator, [B, z, T] like upstream, then to the device
            noise = torch.randn(B, e.z, T).transpose(1, 2)
        return noise

    @torch.no_grad()
    def encode_feature(self, feature, noise=None):
        """feature [B,T,dim] -> posterior sample.  Upstream returns [B,z,T]; so does this (a transposed view)."""
        e = self.engine()
        B, T, _ = feature.shape
        z = e.sample_posterior(e.encode_params(feature), self._posterior_noise(B, T, noise))
        return z.transpose(1, 2)

    def decode_feature(self, latent, mask):
        """latent [B,T,z], mask [B,T] -> (decoded_feature [B,T,dim], lm_result [B,T,1004])."""
        recon, logits, _ = self.engine().decode(latent, _mask_to_lengths(mask), want_units=False)
        return recon, logits

    def forward(self, input_feature, input_token, mask, noise=None, ntokens=None, return_stats=False):
        """-> (mse_loss, lm_result, kl_loss) (:1118-1142).  After `enable_training()` the three are autograd-connected to
        `flat_params` through the HIP backward (`_VaeStepFn`); `return_stats=True` returns the engine's statistics vector
        [loss, nll_loss, mse_loss, kl_loss, acc, ...] instead (the criterion's fused path)."""
        if self._train_engine is not None:
            lengths = _mask_to_lengths(mask)
            if ntokens is None:
                ntokens = int(lengths.sum())
            _set_dropout(self._train_engine, self.attn_dropout if self.training else 0.0)
            if torch.is_grad_enabled():
                stats, logits = _VaeStepFn.apply(self.flat_params, self, input_feature, input_token, lengths, noise, ntokens)
            else:
                stats, logits, _ = self._train_engine.forward(input_feature, input_token, lengths, noise=noise, ntokens=ntokens,
                                                              want_logits=True)
            return (stats, logits) if return_stats else (stats[2], logits, stats[3])
        e = self.engine()
        B, T, _ = input_feature.shape
        lengths = _mask_to_lengths(mask)
        params = e.encode_params(input_feature)
        z, kl = e.sample_posterior(params, self._posterior_noise(B, T, noise), lengths, want_kl=True)
        decoded, logits, _ = e.decode(z, lengths, want_units=False)
        sel = mask.to(decoded.device).unsqueeze(2).expand(-1, -1, decoded.shape[2])
        mse = torch.mean((decoded[sel] - input_feature.to(decoded.device)[sel]) ** 2)
        return mse, logits, kl.mean()


class LatentDiscreteModel(nn.Module):
    """reference latent_module.py:1300-1613.  `speech_decoder` is the fairseq model whose `.encoder` is the VAE."""

    def __init__(self, speech_decoder, dim, latent_dim, target_sample_hz=None, timesteps=1000, use_ddim=True,
                 noise_schedule='sigmoid', objective='v', schedule_kwargs: dict = dict(), time_difference=0.,
                 min_snr_loss_weight=True, min_snr_gamma=5, train_prob_self_cond=0.9, scale=1., use_cond=False,
                 multitask=True, dtype="bf16"):
        super().__init__()
        assert objective in {'x0', 'eps', 'v'}, 'objective must be either predict x0 or noise'
        self.speech_decoder = speech_decoder.encoder
        self.use_cond, self.multitask = use_cond, multitask
        # use_cond: Model(dim, latent_dim, condition_on_prompt=True, dim_prompt=768, num_latents_m=64) upstream (:1325-1333)
        self.model = Model(dim, latent_dim, condition_on_prompt=use_cond, dim_prompt=getattr(speech_decoder.encoder, "dim", 768) if use_cond else None,
                           num_latents_m=64, dtype=dtype)
        self.scheduler = DDPMScheduler(timesteps, scale=scale)
        self.dim, self.timesteps, self.objective = dim, timesteps, objective
        self.min_snr_loss_weight, self.min_snr_gamma = min_snr_loss_weight, min_snr_gamma
        self._coef = None

    @property
    def device(self):
        return self.model.device

    def max_positions(self):
        return None

    _train_engine = None
    train_on_move = False  # set by the plugin's build_model for a training run (see SpeechVAEEncoderDecoder._apply)
    attn_dropout = 0.1  # the eps-predictor's Attention(dropout=0.1) (:668); the frozen VAE stays in eval mode (:1530)

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        if self.train_on_move and self._train_engine is None and not self.use_cond and self.device.type == "cuda":
            self.enable_training()
        return out

    def enable_training(self):
        """Switches the eps-predictor to the HIP diffusion training engine (diffnorm_amd/training.py::EpsTrainEngine): `self.model`
        keeps one flat `flat_params` Parameter aliasing the engine's fp32 master buffer (with `.grad` aliasing its gradient
        buffer); the frozen VAE (diff_discrete.py:79-82) is mirrored into a VaeTrainEngine that only passes data gradients."""
        if self._train_engine is not None:
            return self._train_engine
        if self.use_cond:
            raise NotImplementedError("the HIP training engine covers the unconditional eps-predictor (the recipe); the conditional "
                                      "variant (use_cond) has its forward / guidance path only")
        from . import training

        vae = self.speech_decoder
        dev = self.device
        vsd = {k: v.detach().cpu() for k, v in vae.state_dict().items()}
        self._frozen_vae = training.VaeTrainEngine(vsd, dim=vae.dim, latent_dim=vae.latent_dim, dtype=self.model.arith, device=dev)
        esd = {k: v.detach().cpu() for k, v in self.model.state_dict().items() if not k.startswith("pos_embed")}
        eng = training.EpsTrainEngine(esd, self.model.cfg, self._frozen_vae, timesteps=self.timesteps, dtype=self.model.arith, device=dev,
                                      multitask=self.multitask)
        self.model._adopt_flat(eng)
        self._train_engine = eng
        return eng

    def _tables(self):
        if self._coef is None or self._coef[0].device != self.device:
            s = self.scheduler
            self._coef = (s.ddim_coef_table(self.device), s.f32("sqrt_alphas_cumprod", self.device),
                          s.f32("sqrt_one_minus_alphas_cumprod", self.device))
        return self._coef

    @torch.no_grad()
    def ddim_sample(self, tgt_feature, prompt=None, prompt_mask=None, input_mask=None, cond_scale=1., ref_units=None,
                    start_step=50, post_noise=None, start_noise=None, use_graph=True):
        """-> (list of unit tensors, match, total, recon_feature), as upstream (:1385-1471)."""
        dev = self.device
        coef, sa, s1 = self._tables()
        B, T, _ = tgt_feature.shape
        if input_mask is None:
            input_mask = torch.ones(B, T, dtype=torch.bool, device=dev)
        input_mask = input_mask.to(dev)
        lengths = _mask_to_lengths(input_mask).to(torch.int32)
        z = self.speech_decoder.encode_feature(tgt_feature, noise=post_noise).transpose(1, 2).contiguous()
        if start_noise is None:
            start_noise = torch.randn(z.shape, device=dev)
        t_start = torch.full((B,), start_step, dtype=torch.int32, device=dev)
        x = ops.q_sample(z, start_noise.to(dev, torch.float32).contiguous(), sa, s1, t_start, T)  # (:1405-1409)
        if self.use_cond:
            # prompted chain (f3): the reference's loop drops the prompt it was given (:1413-1417 -- with use_cond it cannot run at
            # all); here every step is the guided prediction forward_with_cond_scale (:813-826) -- conditioned and null rows in ONE pass of
            # twice the batch -- followed by the same DDIM update, one step captured into a hipGraph and replayed (EpsEngine.guided_ddim_chain).
            if prompt is None or prompt_mask is None:
                raise ValueError("use_cond: ddim_sample needs prompt and prompt_mask")
            plens = _mask_to_lengths(prompt_mask.to(dev))
            self.model.engine().guided_ddim_chain(x, lengths, prompt, plens, start_step, coef, cond_scale=cond_scale, use_graph=use_graph)
        else:
            self.model.engine().ddim_loop(x, lengths, start_step, coef, use_graph=use_graph)    # (:1411-1445)
        recon, _, units = self.speech_decoder.engine().decode(x, lengths, want_logits=False)     # (:1448-1451)
        pred_units = units.long()
        match = total = 0
        if ref_units is not None:
            match = (pred_units[input_mask] == ref_units.to(dev)[input_mask]).sum().item()
        total = int(input_mask.sum().item())
        lens = lengths.tolist()
        out_tokens = [pred_units[i, : lens[i]] for i in range(B)]
        return out_tokens, match, total, recon

    @torch.no_grad()
    def ddpm_sample(self, tgt_feature, input_mask=None, ref_units=None, start_step=50, post_noise=None, start_noise=None, seed=0,
                    step_noise=None, fixed_large=False, clip_denoised=False, use_graph=True):
        """The chain of `ddim_sample` with the ancestral (DDPM) update -- GaussianDiffusion.p_sample (reference diffusion/
        gaussian_diffusion.py:376-417) on this model's cosine schedule: encode, noise to index start_step-1 (q_sample), then
        t = start_step-1 .. 0 on the device (dn_ddpm_loop; the per-step noise drawn in the kernel from `seed`, or `step_noise`
        [start_step, B, T, z] injected), decode, units.  BASELINE configs[2] read literally; the reference's LatentDiscreteModel
        itself only has the DDIM sampler.  -> (list of unit tensors, match, total, recon_feature)."""
        dev = self.device
        _, sa, s1 = self._tables()
        B, T, _ = tgt_feature.shape
        if input_mask is None:
            input_mask = torch.ones(B, T, dtype=torch.bool, device=dev)
        input_mask = input_mask.to(dev)
        lengths = _mask_to_lengths(input_mask).to(torch.int32)
        z = self.speech_decoder.encode_feature(tgt_feature, noise=post_noise).transpose(1, 2).contiguous()
        if start_noise is None:
            start_noise = torch.randn(z.shape, device=dev)
        t_start = torch.full((B,), start_step - 1, dtype=torch.int32, device=dev)
        x = ops.q_sample(z, start_noise.to(dev, torch.float32).contiguous(), sa, s1, t_start, T)
        key = ("gd", bool(fixed_large))
        if getattr(self, "_gd_table", (None,))[0] != key or self._gd_table[1].device != dev:
            self._gd_table = (key, self.scheduler.gaussian_table(dev, fixed_large=fixed_large))
        self.model.engine().ddpm_loop(x, lengths, start_step, self._gd_table[1], seed=seed, noise=step_noise, clip_denoised=clip_denoised,
                                      use_graph=use_graph)
        recon, _, units = self.speech_decoder.engine().decode(x, lengths, want_logits=False)
        pred_units = units.long()
        match = (pred_units[input_mask] == ref_units.to(dev)[input_mask]).sum().item() if ref_units is not None else 0
        lens = lengths.tolist()
        return [pred_units[i, : lens[i]] for i in range(B)], match, int(input_mask.sum().item()), recon

    def forward(self, audio, audio_units, src_feature=None, src_mask=None, tgt_mask=None, prompt=None, pitch=None,
                times=None, post_noise=None, jitter_noise=None, true_noise=None, *args, **kwargs):
        """Training loss dict (:1514-1613), forward only.  t, the posterior noise, the beta_0 jitter and the target
        noise can be injected (parity runs); otherwise they are drawn like upstream."""
        dev = self.device
        _, sa_t, s1_t = self._tables()
        B, T, _ = audio.shape
        tgt_mask = tgt_mask.to(dev)
        lengths = _mask_to_lengths(tgt_mask).to(torch.int32)
        if times is None:
            times = torch.randint(1, self.timesteps, (B,), device=dev)  # never 0 (:1528)
        times = times.to(dev)
        if self._train_engine is not None:  # HIP training engine: losses and gradients are kernels
            with torch.no_grad():
                zt = self.speech_decoder.encode_feature(audio, noise=post_noise).transpose(1, 2).contiguous()
            jn = torch.randn(zt.shape, device=dev) if jitter_noise is None else jitter_noise
            tn = torch.randn(zt.shape, device=dev) if true_noise is None else true_noise
            _set_dropout(self._train_engine, self.attn_dropout if self.training else 0.0)
            if torch.is_grad_enabled():
                st = _EpsStepFn.apply(self.model.flat_params, self, audio, audio_units, lengths, zt, times, jn, tn)
            else:
                st = self._train_engine.forward(audio, audio_units, lengths, zt, times, jn, tn)
            return {"total_loss": st[0], "nll_loss": st[1], "recon_mse_loss": st[2], "noise_loss": st[3], "acc": st[4]}
        t32 = times.to(torch.int32)
        z = self.speech_decoder.encode_feature(audio, noise=post_noise).transpose(1, 2).contiguous()
        jitter_noise = torch.randn(z.shape, device=dev) if jitter_noise is None else jitter_noise.to(dev)
        true_noise = torch.randn(z.shape, device=dev) if true_noise is None else true_noise.to(dev, torch.float32).contiguous()
        beta0 = float(self.scheduler.f32("betas")[0])
        x1 = (z + jitter_noise * beta0).contiguous()  # beta_0, not sqrt(beta_0) (:1534-1536)
        xt = ops.q_sample(x1, true_noise, sa_t, s1_t, t32, T)
        if self.use_cond:  # (:1546-1553): the source features are the prompt
            eps = self.model(xt, times, prompt=src_feature, prompt_mask=src_mask, input_mask=tgt_mask, cond_drop_prob=0.1,
                             drop_mask=kwargs.get("drop_mask"))
        else:
            eps = self.model(xt, times, input_mask=tgt_mask, cond_drop_prob=0.1)
        snr = self.scheduler.get_snr(times)
        weight = snr.clamp(max=5.0) / snr
        mse = ((eps - true_noise) ** 2).masked_fill(~tgt_mask.unsqueeze(2), 0.0).flatten(1).mean(dim=1)
        noise_mse = (mse * weight).mean()
        sa = self.scheduler.get_sqrt_alpha_cum(times, xt.shape)
        s1 = self.scheduler.get_sqrt_one_minus_alpha_cum(times, xt.shape)
        x1_hat = ((xt - s1 * eps) / sa.clamp(min=1e-10)).contiguous()
        dec, logits = self.speech_decoder.decode_feature(x1_hat, tgt_mask)
        sel = tgt_mask.unsqueeze(2).expand(-1, -1, dec.shape[2])
        recon_mse = torch.mean((dec[sel] - audio.to(dev)[sel]) ** 2)
        lprobs = torch.log_softmax(logits, dim=-1).view(-1, logits.size(-1))
        unit = audio_units.to(dev).view(-1)
        keep = unit.ne(0)
        acc = torch.sum(lprobs.argmax(1).masked_select(keep).eq(unit.masked_select(keep))) / torch.sum(keep)
        smooth, _ = label_smoothed_nll_loss(lprobs, unit, 0.1, ignore_index=0, reduce=True)
        smooth = smooth / keep.sum()
        recon = 50 * recon_mse + smooth
        total = noise_mse + recon / self.timesteps if self.multitask else noise_mse
        return {"total_loss": total, "nll_loss": smooth, "recon_mse_loss": recon_mse, "noise_loss": noise_mse, "acc": acc}
