"""Training step of the speech VAE on the HIP path (SURVEY 8 f2 + (e)-training, BASELINE config 4): host side.

`VaeTrainEngine` owns the flat device buffers of the C-ABI training engine (csrc/train_engine.hip: master fp32 / work copy /
aux / gradients, all in the packed layout) and exposes forward + staged backward; `GradientReducer` is the one exchange step
of data-parallel training -- the bucketed all-reduce of the flat gradient buffer (RCCL over xGMI through torch.distributed;
the buckets are the ranges the backward stages complete, launched on a side stream as soon as their stage has been enqueued,
so they overlap the rest of the backward pass) plus one small statistics all-reduce; `VaeTrainer.train_step` drives them the
way fairseq's trainer drives a step (fairseq/trainer.py:784-960: forward/backward per micro-batch, gradient reduction,
multiply_grads(world / sample_size), clip_grad_norm, lr schedule, Adam).

PyTorch supplies device memory, streams and torch.distributed; every arithmetic pass is a kernel of libdiffnorm_hip.so.
"""
import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib, optim, packing
from .profiling import profile_range
from .engine import _dtype_code, _require_cuda


def _training_dtype(dtype) -> int:
    """The training engines run exact fp32 (the reference's own training arithmetic: parity at 1e-3) or bf16 (the fast mode);
    f16 and bf16x3 are arithmetic modes of the inference / sampling engines."""
    code = _dtype_code(dtype)
    if code not in (_lib.DN_F32, _lib.DN_BF16):
        raise ValueError(f"training engines run dtype 'f32' or 'bf16'; {dtype!r} is an inference-only arithmetic mode "
                         "(build the model with --hip-dtype bf16 / f32 for a training run)")
    return code


def _aligned_empty(nbytes: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """(owner, 256-byte aligned uint8 view of nbytes)."""
    raw = torch.zeros(nbytes + 256, dtype=torch.uint8, device=device)
    off = (-raw.data_ptr()) % 256
    return raw, raw[off: off + nbytes]


def _dropout_fields(engine):
    """(attn_dropout, seed_lo, seed_hi, pad) of the next forward: `engine.attn_dropout` (0 = eval mode; the reference trains with
    0.1, latent_module.py:668) and a 64-bit counter hash of `engine.dropout_seed` and the number of dropout forwards so far, so
    every micro-batch of every update draws its own mask.  The backward re-derives the mask from the same fields."""
    p = float(getattr(engine, "attn_dropout", 0.0))
    if p <= 0.0:
        return 0.0, 0, 0, 0
    engine._dropout_calls = getattr(engine, "_dropout_calls", 0) + 1
    mixed = (int(getattr(engine, "dropout_seed", 0)) * 0x9E3779B97F4A7C15 + engine._dropout_calls * 0xD1B54A32D192ED03) & ((1 << 64) - 1)
    mixed ^= mixed >> 29
    return p, mixed & 0xFFFFFFFF, (mixed >> 32) & 0xFFFFFFFF, 0


class VaeTrainEngine:
    """SpeechVAEEncoderDecoder training on the GPU (reference latent_module.py:1118-1142 + speech_vae_decoder_loss.py:45-95)."""

    LOSS_WEIGHTS = (0.1, 10.0, 1e-4)  # LS-CE, MSE, KL (speech_vae_decoder_loss.py:80-83)

    def __init__(self, state_dict, dim: int = 768, latent_dim: int = 128, dtype="bf16", device="cuda:0", depth: int = 6,
                 heads: int = 8, dim_head: int = 96, stacks: int = 2, layers: int = 3, vocab: int = 1004):
        self.device = _require_cuda(device)
        self.lib = _lib.load()
        self.dim, self.vocab, self.depth = dim, vocab, depth
        self.mults = packing.vae_mults(latent_dim)
        z = dim
        for m in self.mults:
            z //= m
        self.z = z // 2
        self.dtype = _training_dtype(dtype)
        mults = (C.c_int32 * 4)(*(self.mults + [0] * (4 - len(self.mults))))
        cfg = _lib.VaeConfig(dim, self.z, depth, heads, dim_head, stacks, layers, vocab, len(self.mults), mults, self.dtype)
        self.handle = C.c_void_p()
        _lib.check(self.lib.dn_vae_train_create(C.byref(cfg), C.byref(self.handle)), "dn_vae_train_create")
        self.n_params = int(self.lib.dn_vae_train_param_count(self.handle))
        self.entries = packing.vae_train_entries(dim, self.mults, depth, heads, dim_head, stacks, layers, vocab)
        offs = (C.c_int64 * len(self.entries))()
        n = _lib.check(self.lib.dn_vae_train_offsets(self.handle, offs, len(self.entries)), "dn_vae_train_offsets")
        assert n == len(self.entries), (n, len(self.entries))
        self.offsets = list(offs)
        with torch.cuda.device(self.device):
            self._own = []
            raw, view = _aligned_empty(self.n_params * 4, self.device)
            self._own.append(raw)
            self.master = view.view(torch.float32)
            raw, view = _aligned_empty(self.n_params * 4, self.device)
            self._own.append(raw)
            self.grads = view.view(torch.float32)
            if self.dtype == _lib.DN_BF16:
                raw, view = _aligned_empty(self.n_params * 2, self.device)
                self._own.append(raw)
                self.work = view.view(torch.bfloat16)
            else:
                self.work = self.master
            raw, self.aux = _aligned_empty(int(self.lib.dn_vae_train_aux_bytes(self.handle)), self.device)
            self._own.append(raw)
        _lib.check(self.lib.dn_vae_train_bind(self.handle, self.master.data_ptr(), self.work.data_ptr(), self.aux.data_ptr(),
                                              self.grads.data_ptr()), "dn_vae_train_bind")
        self._ws: Optional[torch.Tensor] = None
        self._batch = None
        self._keep = None
        self.load_state_dict(state_dict)

    def __del__(self):
        if getattr(self, "handle", None) and self.handle.value:
            self.lib.dn_vae_train_destroy(self.handle)
            self.handle = None

    # ---- parameters ----------------------------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        """Reference-layout state dict (keys without the `encoder.` prefix) -> the flat master buffer."""
        flat = packing.pack_flat(sd, self.entries, self.offsets, self.n_params)
        self.master.copy_(flat.to(self.device))
        self.sync_work()

    def sync_work(self):
        """work / aux <- master (after loading or an external update of the master buffer)."""
        if self.work is not self.master:
            self.work.copy_(self.master)  # fp32 -> bf16, round to nearest even (the same rounding dn_adam_step applies)
        self.refresh()

    def refresh(self):
        """aux <- work: call after every optimizer step (dn_adam_step has already written the bf16 work copy)."""
        self.update_count = getattr(self, "update_count", 0) + 1
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dn_vae_train_refresh(self.handle, _lib.current_stream()), "dn_vae_train_refresh")
        self.work_current = True  # (latent_module._prepare_step: cleared by the autograd bridge's backward)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return packing.unpack_flat(self.master, self.entries, self.offsets)

    def grad_dict(self) -> Dict[str, torch.Tensor]:
        """Gradients under the reference's parameter names and shapes."""
        return packing.unpack_flat(self.grads, self.entries, self.offsets)

    def zero_grad(self):
        self.grads.zero_()

    def stage_ranges(self) -> List[Tuple[int, int]]:
        """(offset, count) of the gradient range each backward stage completes, stage 0 first."""
        out = []
        off, cnt = C.c_int64(), C.c_int64()
        for st in range(self.depth + 3):
            _lib.check(self.lib.dn_vae_train_stage_range(self.handle, st, C.byref(off), C.byref(cnt)), "dn_vae_train_stage_range")
            out.append((off.value, cnt.value))
        return out

    @property
    def n_stages(self) -> int:
        return self.depth + 3

    # ---- one step ------------------------------------------------------------------------------------------------------
    def _workspace(self, B: int, T: int):
        need = int(self.lib.dn_vae_train_workspace_bytes(self.handle, B, T))
        if self._ws is None or self._ws.numel() < need + 256:
            self._ws = None
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
        p = self._ws.data_ptr()
        a = (p + 255) & ~255
        return a, self._ws.numel() - (a - p)

    def forward(self, feat: torch.Tensor, units: torch.Tensor, lengths: torch.Tensor, noise: Optional[torch.Tensor] = None,
                ntokens: Optional[int] = None, weights: Sequence[float] = LOSS_WEIGHTS, label_smoothing: float = 0.1,
                loss_scale: float = 1.0, want_logits: bool = False, want_recon: bool = False):
        """feat [B,T,dim] fp32, units [B,T] (dictionary indices, 0 = pad), lengths [B] -> stats fp32 [8] on the device:
        loss, nll_loss, mse_loss, kl_loss, acc, n_valid, lsce/ntokens, 0.  Activations stay in the workspace for `backward`."""
        B, T, _ = feat.shape
        dev = self.device
        feat = feat.to(dev, torch.float32).contiguous()
        units = units.to(dev, torch.int32).contiguous()
        lengths = lengths.to(dev, torch.int32).contiguous()
        if noise is None:  # CPU generator, [B, z, T] like upstream (distributions.py:38-40)
            noise = torch.randn(B, self.z, T).transpose(1, 2)
        elif isinstance(noise, tuple):  # ("philox", seed, offset): the build's own device generator (throughput runs)
            from . import ops

            with torch.cuda.device(dev):
                noise = ops.randn((B, T, self.z), seed=int(noise[1]), offset=int(noise[2]), device=dev)
        noise = noise.to(dev, torch.float32).contiguous()
        if ntokens is None:
            ntokens = int(lengths.sum().item())
        stats = torch.empty(8, dtype=torch.float32, device=dev)
        logits = torch.empty(B, T, self.vocab, dtype=torch.float32, device=dev) if want_logits else None
        recon = torch.empty(B, T, self.dim, dtype=torch.float32, device=dev) if want_recon else None
        b = _lib.VaeTrainBatch(feat.data_ptr(), units.data_ptr(), lengths.data_ptr(), noise.data_ptr(), B, T, int(ntokens),
                               float(weights[0]), float(weights[1]), float(weights[2]), float(label_smoothing), float(loss_scale),
                               stats.data_ptr(), _lib.ptr(logits), _lib.ptr(recon), None, *_dropout_fields(self))
        self._batch, self._keep = b, (feat, units, lengths, noise, stats, logits, recon)
        wp, wn = self._workspace(B, T)
        with torch.cuda.device(dev):
            _lib.check(self.lib.dn_vae_train_forward(self.handle, C.byref(b), wp, wn, _lib.current_stream()), "dn_vae_train_forward")
        return (stats, logits, recon) if (want_logits or want_recon) else stats

    def backward(self, first_stage: int = 0, last_stage: Optional[int] = None, ext_dlogits: Optional[torch.Tensor] = None,
                 d_mse: Optional[float] = None, d_kl: Optional[float] = None):
        """Backward stages of the last `forward`; gradients are added to `self.grads`.  `ext_dlogits` [B,T,vocab] (with `d_mse`,
        `d_kl` = d loss / d mse_loss, d loss / d kl_loss) replaces the fused criterion gradient: the path of a caller that
        differentiates (mse_loss, logits, kl_loss) itself."""
        assert self._batch is not None, "backward() needs a forward() first"
        if ext_dlogits is not None:
            ext_dlogits = ext_dlogits.to(self.device, torch.float32).contiguous()
            self._ext = ext_dlogits
            self._batch.ext_dlogits = ext_dlogits.data_ptr()
            self._batch.w_mse, self._batch.w_kl = float(d_mse), float(d_kl)
        last_stage = self.n_stages - 1 if last_stage is None else last_stage
        B, T = self._batch.B, self._batch.T
        wp, wn = self._workspace(B, T)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dn_vae_train_backward(self.handle, C.byref(self._batch), first_stage, last_stage, wp, wn,
                                                      _lib.current_stream()), "dn_vae_train_backward")


class _FlatEngine:
    """Flat master / work / aux / gradient buffers of a training engine and the state-dict conversions over its entry table."""

    def _alloc(self, aux_bytes: int):
        with torch.cuda.device(self.device):
            self._own = []
            raw, view = _aligned_empty(self.n_params * 4, self.device)
            self._own.append(raw)
            self.master = view.view(torch.float32)
            raw, view = _aligned_empty(self.n_params * 4, self.device)
            self._own.append(raw)
            self.grads = view.view(torch.float32)
            if self.dtype == _lib.DN_BF16:
                raw, view = _aligned_empty(self.n_params * 2, self.device)
                self._own.append(raw)
                self.work = view.view(torch.bfloat16)
            else:
                self.work = self.master
            raw, self.aux = _aligned_empty(aux_bytes, self.device)
            self._own.append(raw)

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        flat = packing.pack_flat(sd, self.entries, self.offsets, self.n_params)
        self.master.copy_(flat.to(self.device))
        self.sync_work()

    def sync_work(self):
        if self.work is not self.master:
            self.work.copy_(self.master)
        self.refresh()

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return packing.unpack_flat(self.master, self.entries, self.offsets)

    def grad_dict(self) -> Dict[str, torch.Tensor]:
        return packing.unpack_flat(self.grads, self.entries, self.offsets)

    def zero_grad(self):
        self.grads.zero_()

    def _ws_ptr(self, need: int):
        if self._ws is None or self._ws.numel() < need + 256:
            self._ws = None
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
        p = self._ws.data_ptr()
        a = (p + 255) & ~255
        return a, self._ws.numel() - (a - p)


class EpsTrainEngine(_FlatEngine):
    """The diffusion training step on the GPU (reference LatentDiscreteModel.forward latent_module.py:1514-1613): eps-predictor
    forward / backward with FiLM and adaptive-norm time conditioning, min-SNR noise loss, and the multitask reconstruction losses
    through the FROZEN VAE (`vae`: a VaeTrainEngine whose parameters receive no gradient, diff_discrete.py:79-82)."""

    def __init__(self, state_dict, cfg, vae: Optional["VaeTrainEngine"], timesteps: int = 200, dtype="bf16", device="cuda:0",
                 max_pos: int = 2048, multitask: bool = True):
        from . import scheduler

        self.device = _require_cuda(device)
        self.lib = _lib.load()
        self.cfg, self.vae, self.timesteps, self.multitask = cfg, vae, timesteps, multitask
        self.dtype = _training_dtype(dtype)
        self.depth = cfg.depth
        c = _lib.EpsConfig(cfg.dim, cfg.latent_dim, cfg.depth, cfg.heads, cfg.dim_head, cfg.wavenet_layers, cfg.wavenet_stacks,
                           cfg.dim_cond_mult, self.dtype, max_pos)
        self.handle = C.c_void_p()
        _lib.check(self.lib.dn_eps_train_create(C.byref(c), C.byref(self.handle)), "dn_eps_train_create")
        self.n_params = int(self.lib.dn_eps_train_param_count(self.handle))
        self.entries = packing.eps_train_entries(cfg)
        offs = (C.c_int64 * len(self.entries))()
        n = _lib.check(self.lib.dn_eps_train_offsets(self.handle, offs, len(self.entries)), "dn_eps_train_offsets")
        assert n == len(self.entries), (n, len(self.entries))
        self.offsets = list(offs)
        self._alloc(int(self.lib.dn_eps_train_aux_bytes(self.handle)))
        raw, view = _aligned_empty((max_pos + 1) * packing.padk(cfg.dim) * 4, self.device)
        self._own.append(raw)
        self.pos_table = view.view(torch.float32)
        self.pos_table.copy_(packing.sinusoidal_table(max_pos + 1, cfg.dim, packing.padk(cfg.dim)).reshape(-1))
        _lib.check(self.lib.dn_eps_train_bind(self.handle, self.master.data_ptr(), self.work.data_ptr(), self.aux.data_ptr(),
                                              self.grads.data_ptr(), self.pos_table.data_ptr()), "dn_eps_train_bind")
        self.sched = scheduler.DDPMScheduler(timesteps)
        self._sa = self.sched.f32("sqrt_alphas_cumprod", self.device)
        self._s1 = self.sched.f32("sqrt_one_minus_alphas_cumprod", self.device)
        self._beta0 = float(self.sched.f32("betas")[0])
        self._ws = None
        self._batch = None
        self.load_state_dict(state_dict)

    def __del__(self):
        if getattr(self, "handle", None) and self.handle.value:
            self.lib.dn_eps_train_destroy(self.handle)
            self.handle = None

    def refresh(self):
        self.update_count = getattr(self, "update_count", 0) + 1
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dn_eps_train_refresh(self.handle, _lib.current_stream()), "dn_eps_train_refresh")
        self.work_current = True

    @property
    def n_stages(self) -> int:
        return self.depth + 3

    def stage_ranges(self) -> List[Tuple[int, int]]:
        out = []
        off, cnt = C.c_int64(), C.c_int64()
        for st in range(self.n_stages):
            _lib.check(self.lib.dn_eps_train_stage_range(self.handle, st, C.byref(off), C.byref(cnt)), "dn_eps_train_stage_range")
            out.append((off.value, cnt.value))
        return out

    def _vae_handle(self):
        return self.vae.handle if (self.vae is not None and self.multitask) else None

    def forward(self, feat, units, lengths, z, times, jitter, true_noise, loss_scale: float = 1.0, want_eps: bool = False,
                n_units: Optional[int] = None, n_frames: Optional[int] = None):
        """-> stats fp32 [8]: total_loss, nll_loss, recon_mse_loss, noise_loss, acc, n_units.  z [B,T,latent]: the frozen encoder's
        posterior sample; times [B] in [1, timesteps); jitter / true_noise [B,T,latent].  n_units (non-pad units) / n_frames (sum of
        the lengths) are host scalars of the loss normalisation: pass them (the collater knows both) or they are read back from the
        device tensors here -- a stream synchronisation per update that keeps the host from enqueuing the next update behind this one."""
        dev = self.device
        B, T, _ = z.shape
        f32 = lambda t: t.to(dev, torch.float32).contiguous()
        feat, z, jitter, true_noise = f32(feat), f32(z), f32(jitter), f32(true_noise)
        units = units.to(dev, torch.int32).contiguous()
        lengths = lengths.to(dev, torch.int32).contiguous()
        times = times.to(dev, torch.int32).contiguous()
        snr = self.sched.get_snr(times.long())
        weight = (snr.clamp(max=5.0) / snr).to(dev, torch.float32).contiguous()  # min-SNR-5 (:1565-1569)
        n_units = int((units != 0).sum().item()) if n_units is None else int(n_units)
        n_frames = int(lengths.sum().item()) if n_frames is None else int(n_frames)
        stats = torch.empty(8, dtype=torch.float32, device=dev)
        eps = torch.empty(B, T, self.cfg.latent_dim, dtype=torch.float32, device=dev) if want_eps else None
        b = _lib.EpsTrainBatch(feat.data_ptr(), units.data_ptr(), lengths.data_ptr(), z.data_ptr(), jitter.data_ptr(), true_noise.data_ptr(),
                               times.data_ptr(), self._sa.data_ptr(), self._s1.data_ptr(), weight.data_ptr(), self._beta0, B, T, n_units,
                               n_frames, self.timesteps, int(self.multitask), 0.1, 50.0, float(loss_scale), stats.data_ptr(), _lib.ptr(eps),
                               *_dropout_fields(self))
        self._batch, self._keep = b, (feat, units, lengths, z, jitter, true_noise, times, weight, stats, eps)
        need = int(self.lib.dn_eps_train_workspace_bytes(self.handle, self._vae_handle(), B, T))
        wp, wn = self._ws_ptr(need)
        with torch.cuda.device(dev):
            _lib.check(self.lib.dn_eps_train_forward(self.handle, self._vae_handle(), C.byref(b), wp, wn, _lib.current_stream()),
                       "dn_eps_train_forward")
        return (stats, eps) if want_eps else stats

    def backward(self, first_stage: int = 0, last_stage: Optional[int] = None):
        assert self._batch is not None, "backward() needs a forward() first"
        last_stage = self.n_stages - 1 if last_stage is None else last_stage
        need = int(self.lib.dn_eps_train_workspace_bytes(self.handle, self._vae_handle(), self._batch.B, self._batch.T))
        wp, wn = self._ws_ptr(need)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dn_eps_train_backward(self.handle, self._vae_handle(), C.byref(self._batch), first_stage, last_stage, wp, wn,
                                                      _lib.current_stream()), "dn_eps_train_backward")


def plan_buckets(ranges: Sequence[Tuple[int, int]], min_elems: int) -> List[Tuple[int, int, int]]:
    """Merges consecutive backward-stage ranges (each ends where the previous one starts: the backward walks the buffer from
    its end) into buckets of at least `min_elems` elements.  -> [(last_stage_of_bucket, offset, count)], in completion order."""
    buckets, lo, n, prev_lo = [], None, 0, None
    for stage, (off, cnt) in enumerate(ranges):
        if cnt == 0:
            continue
        assert prev_lo is None or off + cnt == prev_lo, "stage ranges must be contiguous, descending"
        prev_lo = lo = off
        n += cnt
        if n >= min_elems:
            buckets.append((stage, lo, n))
            n = 0
    if n:
        buckets.append((len(ranges) - 1, lo, n))
    return buckets


class GradientReducer:
    """Bucketed sum-all-reduce of a flat gradient buffer, overlapped with the backward pass.

    Replaces the DDP reducer of the reference's training stack (fairseq/models/distributed_fairseq_model.py:59-84: torch DDP with
    25 MB buckets; fairseq/trainer.py:912-916).  One process per GPU; backend "nccl" is RCCL on ROCm.  xGMI is point-to-point, a
    ring all-reduce is bound by one link, so the buckets are few and large (the ranges the backward stages complete, merged up to
    `bucket_mb`); each is issued on a side stream right after its stage has been enqueued on the compute stream.
    On CPU tensors (gloo; the CPU tests) the same bucket plan runs synchronously."""

    def __init__(self, grads: torch.Tensor, ranges: Sequence[Tuple[int, int]], group=None, bucket_mb: float = 64.0):
        import torch.distributed as dist

        self.dist = dist
        self.grads, self.group = grads, group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.buckets = plan_buckets(ranges, int(bucket_mb * (1 << 20) / 4))
        self.cuda = grads.is_cuda
        self.comm_stream = torch.cuda.Stream(device=grads.device) if self.cuda else None
        self._handles = []
        self._timing: List[Tuple[torch.cuda.Event, torch.cuda.Event]] = []
        self.measure = False

    def bucket_after_stage(self, stage: int) -> Optional[int]:
        for i, (last, _, _) in enumerate(self.buckets):
            if last == stage:
                return i
        return None

    def reduce_bucket(self, i: int):
        """Call after the last stage of bucket i has been enqueued on the current stream."""
        if self.world == 1:
            return
        _, off, cnt = self.buckets[i]
        view = self.grads[off: off + cnt]
        if not self.cuda:
            self.dist.all_reduce(view, group=self.group)
            return
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(ready)
            if self.measure:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            self._handles.append(self.dist.all_reduce(view, group=self.group, async_op=True))
            if self.measure:
                self._handles[-1].wait()
                e1.record()
                self._timing.append((e0, e1))

    def finish(self):
        """The compute stream waits for every outstanding bucket."""
        if self.world == 1:
            return
        for h in self._handles:
            h.wait()
        self._handles = []
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def all_reduce_ms(self) -> float:
        """Sum of the measured bucket all-reduce times since the last call (needs `measure = True`; synchronises)."""
        if not self._timing:
            return 0.0
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self._timing)
        self._timing = []
        return ms


class VaeTrainer:
    """One process's half of fairseq's Trainer.train_step (fairseq/trainer.py:784-960) for --task speech_decoder --criterion
    speech_vae_decoder_loss on the HIP engine: per micro-batch forward + backward, gradient all-reduce (sum), the statistics
    all-reduce, multiply_grads(world / sample_size) -- here 1 / sample_size on the summed gradient --, clip_grad_norm, the
    inverse_sqrt learning rate of update `num_updates`, Adam.  No host synchronisation inside a step."""

    def __init__(self, engine: VaeTrainEngine, lr: float = 5e-4, betas=(0.9, 0.98), eps: float = 1e-8, weight_decay: float = 0.0,
                 clip_norm: float = 2.0, warmup_updates: int = 10000, warmup_init_lr: float = 1e-7, group=None,
                 bucket_mb: float = 64.0, adam=None, attn_dropout: float = 0.1, seed: int = 1):
        self.engine = engine
        self.attn_dropout, self.seed = float(attn_dropout), int(seed)  # train mode of the reference: Attention(dropout=0.1)
        # `adam`: anything with set_lr / step(grads, grad_scale, grad_scale_dev) -- the CPU tests of the exchange logic pass a recorder
        self.adam = adam if adam is not None else optim.Adam(
            engine.master, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, clip_norm=clip_norm,
            bf16_copy=engine.work if engine.work is not engine.master else None)
        self.schedule = optim.InverseSquareRootSchedule(lr, warmup_updates, warmup_init_lr)
        self.reducer = GradientReducer(engine.grads, engine.stage_ranges(), group=group, bucket_mb=bucket_mb)
        self.group = group
        self.num_updates = 0

    def _forward(self, sample: dict, noise):
        return self.engine.forward(sample["reduce_target"], sample["reduce_target_unit"], sample["reduce_target_lengths"], noise=noise,
                                   ntokens=int(sample["ntokens"]))

    def train_step(self, samples: Sequence[dict], noises: Optional[Sequence[torch.Tensor]] = None):
        """samples: the micro-batches of one update (update_freq), each the criterion's sample dict.
        -> (stats [8] of the last micro-batch weighted like the criterion's logging output, grad_norm) as device tensors."""
        eng, red = self.engine, self.reducer
        eng.attn_dropout, eng.dropout_seed = self.attn_dropout, self.seed + self.num_updates  # fairseq seeds an update with seed + num_updates
        eng.zero_grad()
        totals = torch.zeros(10, dtype=torch.float32, device=eng.device)  # sum_i nsent_i * stats_i [0:8], nsentences, ntokens
        for k, sample in enumerate(samples):
            with profile_range("forward"):
                stats = self._forward(sample, None if noises is None else noises[k])
            nsent = float(sample["nsentences"])
            totals[:8] += stats * nsent
            totals[8] += nsent
            totals[9] += float(sample["ntokens"])
            with profile_range("backward"):
                if k + 1 < len(samples):
                    eng.backward()
                    continue
                for stage in range(eng.n_stages):  # last micro-batch: ranges enter the all-reduce as they complete
                    eng.backward(stage, stage)
                    b = red.bucket_after_stage(stage)
                    if b is not None:
                        red.reduce_bucket(b)
        with profile_range("reduce-grads"):  # (the buckets were issued under the backward; this is the wait + the statistics exchange)
            if red.world > 1:
                red.dist.all_reduce(totals, group=self.group)  # the one small statistics exchange (replaces all_gather_list of dicts)
            red.finish()
        inv_sample_size = 1.0 / totals[8:9]  # sample_size = sum of nsentences over ranks and micro-batches (criterion :84)
        self.adam.set_lr(self.schedule.step_update(self.num_updates))
        with profile_range("optimizer"):  # multiply-grads and clip-grads are not passes of their own: both happen inside dn_adam_step
            grad_norm = self.adam.step(eng.grads, grad_scale=1.0, grad_scale_dev=inv_sample_size)
            eng.refresh()
        self.num_updates += 1
        logged = totals[:8] / totals[8]  # sample-size-weighted means, as reduce_metrics (:97-112)
        return logged, grad_norm


class DiffusionTrainer(VaeTrainer):
    """The same update loop for --task speech_diffusion_discrete --criterion ddpm_discrete_loss (scripts/diffusion/train.sh:
    lr 1e-4, Adam (0.9, 0.98), clip-norm 2.0, inverse_sqrt): `ldm` is the mirror LatentDiscreteModel; its frozen VAE encodes the
    features (no gradient), the HIP diffusion engine does the rest.  `noises[k]` (optional) = dict(times, post_noise, jitter_noise,
    true_noise) injected for micro-batch k; otherwise t ~ U{1..T-1} and every noise tensor drawn ON THE DEVICE.  (The reference draws
    the posterior noise with the CPU generator and copies it over, distributions.py:41 -- a million normals and a pageable H2D copy
    per update, host work that showed as 31 vs 36-39 ms per update between boxes; `reference_rng=True` keeps that behaviour.)"""

    def __init__(self, ldm, lr: float = 1e-4, reference_rng: bool = False, **kw):
        self.ldm = ldm
        self.reference_rng = reference_rng
        super().__init__(ldm.enable_training(), lr=lr, **kw)

    def _forward(self, sample: dict, draws):
        draws = draws or {}
        ldm, eng = self.ldm, self.engine
        feat, lens = sample["reduce_target"], sample["reduce_target_lengths"]
        B = feat.shape[0]
        times = draws.get("times")
        if times is None:
            times = torch.randint(1, ldm.timesteps, (B,), device=eng.device)
        # the three noise tensors come from the library's own Philox4x32-10 kernel (dn_randn), keyed by (seed + num_updates, micro-batch,
        # which tensor): no torch RNG kernel in the step, an update is reproducible from the trainer's seed alone
        from . import ops

        self._draws = getattr(self, "_draws", 0) + 1
        key = lambda i: ((self.seed + self.num_updates) << 20) + ((self._draws & 0xffff) << 4) + i
        post = draws.get("post_noise")
        if post is None and not self.reference_rng:
            post = ops.randn((B, feat.shape[1], ldm.speech_decoder.engine().z), seed=key(0), device=eng.device)
        with torch.no_grad():
            z = ldm.speech_decoder.encode_feature(feat, noise=post).transpose(1, 2).contiguous()
        jn = draws.get("jitter_noise")
        tn = draws.get("true_noise")
        jn = ops.randn(tuple(z.shape), seed=key(1), device=eng.device) if jn is None else jn
        tn = ops.randn(tuple(z.shape), seed=key(2), device=eng.device) if tn is None else tn
        # host scalars of the loss normalisation without a device read-back: the sample dict carries ntokens (= the sum of the
        # lengths) and, from this build's collaters, n_units; an unknown batch is counted once and remembered
        if "n_units" not in sample:
            sample["n_units"] = int((sample["reduce_target_unit"] != 0).sum())
        return eng.forward(feat, sample["reduce_target_unit"], lens, z, times, jn, tn, n_units=sample["n_units"], n_frames=int(sample["ntokens"]))
