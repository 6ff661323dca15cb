"""Thin Python handles over the C-ABI engine objects (DnEps / DnVae) of libdiffnorm_hip.so.

PyTorch is used only for device memory (packed weights, workspaces, I/O tensors) and the current
stream; all arithmetic runs in the HIP library.  There is no CPU path: constructing an engine
without a GPU or without the library raises.
"""
import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib, packing


def _dtype_code(dtype) -> int:
    if isinstance(dtype, str):
        names = {"bf16": _lib.DN_BF16, "f32": _lib.DN_F32, "fp32": _lib.DN_F32, "bf16x3": _lib.DN_BF16X3, "f16": _lib.DN_F16, "fp16": _lib.DN_F16}
        if dtype in names:
            return names[dtype]
    elif dtype is torch.bfloat16:
        return _lib.DN_BF16
    elif dtype is torch.float32:
        return _lib.DN_F32
    elif dtype is torch.float16:
        return _lib.DN_F16
    elif isinstance(dtype, int) and dtype in (_lib.DN_F32, _lib.DN_BF16, _lib.DN_BF16X3, _lib.DN_F16):
        return dtype
    raise ValueError(f"unsupported arithmetic dtype {dtype!r} (use 'bf16', 'f16', 'bf16x3' or 'f32')")


def _require_cuda(device) -> torch.device:
    device = torch.device(device)
    if device.type != "cuda" or not torch.cuda.is_available():
        raise _lib.DiffNormHipError("diffnorm_amd engines need a HIP device (cuda:N); there is no CPU fallback")
    return device


def _i32(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(device=device, dtype=torch.int32).contiguous()


def _f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(device=device, dtype=torch.float32).contiguous()


class _Engine:
    def __init__(self, device, tensors: List[torch.Tensor]):
        self.device = _require_cuda(device)
        self.lib = _lib.load()
        self.tensors = [t.to(self.device).contiguous() for t in tensors]  # keeps packed weights alive
        self._table = (C.c_void_p * len(self.tensors))(*[t.data_ptr() for t in self.tensors])
        self._ws: Optional[torch.Tensor] = None
        self.handle = C.c_void_p()

    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        return self._ws

    @staticmethod
    def _aligned(ws: torch.Tensor):
        p = ws.data_ptr()
        a = (p + 255) & ~255
        return a, ws.numel() - (a - p)

    def weight_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.tensors)


class EpsEngine(_Engine):
    """eps-predictor `Model` (reference latent_module.py:709-876) on the GPU."""

    def __init__(self, state_dict, cfg, dtype="bf16", device="cuda:0", max_pos: int = 2048):
        self.cfg = cfg
        self.dtype = _dtype_code(dtype)
        super().__init__(device, packing.pack_eps(state_dict, cfg, self.dtype, max_pos))
        self.conditional = getattr(cfg, "dim_prompt", 0) > 0
        c = _lib.EpsConfig(cfg.dim, cfg.latent_dim, cfg.depth, cfg.heads, cfg.dim_head, cfg.wavenet_layers,
                           cfg.wavenet_stacks, cfg.dim_cond_mult, self.dtype, max_pos, getattr(cfg, "dim_prompt", 0),
                           getattr(cfg, "num_latents_m", 0) if self.conditional else 0,
                           getattr(cfg, "resampler_depth", 0) if self.conditional else 0)
        _lib.check(self.lib.dn_eps_create(C.byref(c), self._table, len(self.tensors), C.byref(self.handle)),
                   "dn_eps_create")

    def __del__(self):
        if getattr(self, "handle", None) and self.handle.value:
            self.lib.dn_eps_destroy(self.handle)
            self.handle = None

    def workspace_bytes(self, B: int, T: int) -> int:
        return int(self.lib.dn_eps_workspace_bytes(self.handle, B, T))

    def forward(self, x: torch.Tensor, times: torch.Tensor, lengths: torch.Tensor, shared_t: bool = False,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x [B,T,z] fp32, times [B] int, lengths [B] int -> eps_hat [B,T,z] fp32 (Model.forward :828-876)."""
        B, T, z = x.shape
        assert z == self.cfg.latent_dim
        x = _f32(x, self.device)
        t32, l32 = _i32(times, self.device), _i32(lengths, self.device)
        out = torch.empty_like(x) if out is None else out
        ws = self._workspace(self.workspace_bytes(B, T))
        wp, wn = self._aligned(ws)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dn_eps_forward(self.handle, x.data_ptr(), t32.data_ptr(), l32.data_ptr(), B, T,
                                               int(shared_t), out.data_ptr(), wp, wn, _lib.current_stream()),
                       "dn_eps_forward")
        return out

    def cond_time_table(self, t0: int, n_t: int) -> torch.Tensor:
        """The time half of the conditioning rows of timesteps t0 .. t0 + n_t - 1 (dn_eps_cond_time_table): fp32 [n_t, n_cond]."""
        n_cond = (self.cfg.wavenet_stacks * self.cfg.wavenet_layers + 3 * self.cfg.depth) * 2 * packing.padk(self.cfg.dim)
        table = torch.empty(n_t, n_cond, dtype=torch.float32, device=self.device)
        ws = torch.empty(int(self.lib.dn_eps_cond_time_table_workspace_bytes(self.handle, n_t)) + 256, dtype=torch.uint8, device=self.device)
        wp, wn = self._aligned(ws)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dn_eps_cond_time_table(self.handle, int(t0), int(n_t), table.data_ptr(), wp, wn, _lib.current_stream()),
                       "dn_eps_cond_time_table")
        return table

    def forward_cond(self, x: torch.Tensor, times: torch.Tensor, lengths: torch.Tensor, prompt: torch.Tensor, prompt_lengths: torch.Tensor,
                     drop: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, reuse_prompt: bool = False,
                     time_table: Optional[torch.Tensor] = None, table_t0: int = 0) -> torch.Tensor:
        """Conditional variant (reference Model.forward with condition_on_prompt, latent_module.py:828-876): prompt [B,Tp,dim_prompt],
        prompt_lengths [B]; drop [B] bool = the classifier-free-guidance drop mask (True: null condition).  Inside a chain:
        `reuse_prompt` skips the prompt-only work a previous call left in the workspace (same shapes, same prompt and drop mask),
        `time_table` (cond_time_table) supplies the time half of the conditioning rows."""
        B, T, z = x.shape
        Tp = prompt.shape[1]
        x, prompt = _f32(x, self.device), _f32(prompt, self.device)
        t32, l32, p32 = _i32(times, self.device), _i32(lengths, self.device), _i32(prompt_lengths, self.device)
        d32 = torch.zeros(B, dtype=torch.int32, device=self.device) if drop is None else _i32(drop, self.device)
        out = torch.empty_like(x) if out is None else out
        ws = self._workspace(int(self.lib.dn_eps_cond_workspace_bytes(self.handle, B, T, Tp)))
        wp, wn = self._aligned(ws)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dn_eps_forward_cond_ex(self.handle, x.data_ptr(), t32.data_ptr(), l32.data_ptr(), prompt.data_ptr(), p32.data_ptr(),
                                                       d32.data_ptr(), B, T, Tp, out.data_ptr(), wp, wn, int(bool(reuse_prompt)),
                                                       _lib.ptr(time_table), int(table_t0), 0 if time_table is None else time_table.shape[0],
                                                       _lib.current_stream()), "dn_eps_forward_cond")
        return out

    def _guided_inputs(self, lengths, prompt, prompt_lengths):
        """Static inputs of a guided pass over 2B rows = [conditioned ; null-conditioned] (the drop mask is per sample, :843-859)."""
        B = prompt.shape[0]
        two = lambda t: torch.cat([t, t]).contiguous()
        drop2 = torch.cat([torch.zeros(B, dtype=torch.int32), torch.ones(B, dtype=torch.int32)]).to(self.device)
        return two(_i32(lengths, self.device)), two(_f32(prompt, self.device)), two(_i32(prompt_lengths, self.device)), drop2

    def forward_with_cond_scale(self, x, times, lengths, prompt, prompt_lengths, cond_scale: float = 1.0) -> torch.Tensor:
        """Classifier-free guidance (latent_module.py:813-826): null + (cond - null) * cond_scale.  The conditioned and the null pass
        are ONE launch sequence over 2B rows (the engine's drop mask is per sample); a scale of 1 needs the conditioned rows only."""
        B = x.shape[0]
        if cond_scale == 1.0:
            return self.forward_cond(x, times, lengths, prompt, prompt_lengths, torch.zeros(B, dtype=torch.bool))
        x = _f32(x, self.device)
        l2, p2, pl2, drop2 = self._guided_inputs(lengths, prompt, prompt_lengths)
        t2 = torch.cat([_i32(times, self.device)] * 2)
        both = self.forward_cond(torch.cat([x, x]).contiguous(), t2, l2, p2, pl2, drop2)
        out = torch.empty_like(x)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dn_cfg_combine(both.data_ptr(), float(cond_scale), x.numel(), out.data_ptr(), _lib.current_stream()), "dn_cfg_combine")
        return out

    def guided_ddim_chain(self, x: torch.Tensor, lengths, prompt, prompt_lengths, start_step: int, coef: torch.Tensor, cond_scale: float = 1.0,
                          use_graph: bool = True, hoist: bool = True) -> int:
        """The prompted chain of the conditional variant, in place on x [B,T,z]: for t = start_step-1 .. 1 (t = 0 only when start_step
        == 1, like the reference's loop :1411-1445) the guided prediction (one 2B-row pass, or B rows at scale 1) and the DDIM eta = 0
        update.  One step -- timestep fill from a device counter, the pass, the guidance combination, the update, the decrement -- is
        captured into a hipGraph (torch.cuda.CUDAGraph over this library's launches on the capture stream) and replayed: the host only
        issues graph launches.  Returns the number of model evaluations."""
        from . import ops

        B, T, z = x.shape
        assert x.is_contiguous() and x.dtype == torch.float32 and x.device == self.device
        guided = cond_scale != 1.0
        n = 2 * B if guided else B
        if guided:
            l2, p2, pl2, drop2 = self._guided_inputs(lengths, prompt, prompt_lengths)
        else:
            l2, p2, pl2 = _i32(lengths, self.device), _f32(prompt, self.device), _i32(prompt_lengths, self.device)
            drop2 = torch.zeros(B, dtype=torch.int32, device=self.device)
        tvec = torch.full((n,), start_step - 1, dtype=torch.int32, device=self.device)
        xin = torch.empty(n, T, z, dtype=torch.float32, device=self.device) if guided else x
        both = torch.empty(n, T, z, dtype=torch.float32, device=self.device)
        eps = torch.empty_like(x) if guided else both

        # prompt-only work (pooled-prompt half of the conditioning projection, resampler, every layer's prompt keys / values) once per
        # chain: the first step computes it into the workspace, every later step reuses it; the time half of the conditioning rows of
        # all the chain's steps comes from one table
        table = self.cond_time_table(0, start_step) if hoist else None
        first = [True]

        def one_step():
            if guided:
                xin[:B].copy_(x)
                xin[B:].copy_(x)
            self.forward_cond(xin, tvec, l2, p2, pl2, drop2, out=both, reuse_prompt=hoist and not first[0], time_table=table, table_t0=0)
            first[0] = False
            if guided:
                _lib.check(self.lib.dn_cfg_combine(both.data_ptr(), float(cond_scale), x.numel(), eps.data_ptr(), _lib.current_stream()), "dn_cfg_combine")
            ops.ddim_step(x, eps, coef, tvec[:B], T, out=x)
            tvec.sub_(1)

        n_eval = max(1, start_step - 1)
        with torch.cuda.device(self.device):
            one_step()  # eager first step: settles workspaces and kernel attributes outside capture
            done = 1
            if use_graph and n_eval > 2:
                cur = torch.cuda.current_stream()
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(cur)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(side):
                    with torch.cuda.graph(g, stream=side):
                        one_step()
                    done += 0  # (capture does not execute)
                    for _ in range(n_eval - done):
                        g.replay()
                cur.wait_stream(side)
                done = n_eval
            for _ in range(n_eval - done):
                one_step()
        return n_eval

    def ddim_loop(self, x: torch.Tensor, lengths: torch.Tensor, start_step: int, coef: torch.Tensor,
                  use_graph: bool = True, max_evals: int = 0, split: bool = True, keep_table: bool = False) -> int:
        """In-place DDIM eta=0 chain on x [B,T,z] fp32 (reference latent_module.py:1411-1445).
        coef: fp32 [timesteps,4] from `scheduler.ddim_coef_table`.  Returns the number of model evaluations.
        `max_evals` stops early; the caller continues with start_step - max_evals and may pass `keep_table=True` when
        nothing else used this engine in between (the conditioning table of the first call is then reused)."""
        B, T, z = x.shape
        assert x.is_contiguous() and x.dtype == torch.float32 and x.device == self.device
        assert coef.dtype == torch.float32 and coef.is_contiguous() and coef.device == self.device
        l32 = lengths if (lengths.dtype == torch.int32 and lengths.device == self.device) else _i32(lengths, self.device)
        self._keep = (l32, coef)
        ws = self._workspace(int(self.lib.dn_ddim_workspace_bytes(self.handle, B, T, start_step)))
        wp, wn = self._aligned(ws)
        with torch.cuda.device(self.device):
            return _lib.check(self.lib.dn_ddim_loop(self.handle, x.data_ptr(), l32.data_ptr(), B, T, start_step,
                                                    max_evals, coef.data_ptr(), coef.shape[0],
                                                    (1 if use_graph else 0) | (2 if split else 0) | (4 if keep_table else 0), wp, wn,
                                                    _lib.current_stream()), "dn_ddim_loop")


    def ddpm_loop(self, x: torch.Tensor, lengths: torch.Tensor, start_step: int, table: torch.Tensor, seed: int = 0,
                  noise: Optional[torch.Tensor] = None, clip_denoised: bool = False, use_graph: bool = True, max_evals: int = 0,
                  split: bool = True, keep_table: bool = False) -> int:
        """In-place ancestral (DDPM) chain on x [B,T,z] fp32: GaussianDiffusion.p_sample (reference diffusion/gaussian_diffusion.py:
        376-417) for t = start_step-1 .. 0, x given at index start_step-1 (x_T ~ N(0, I) with start_step = timesteps: BASELINE
        configs[2] read literally).  table: fp32 [timesteps, 12] from `scheduler.gaussian_table`.  The noise of a step is drawn
        in the update kernel (Philox keyed by `seed` and the step) or injected: noise [start_step, B, T, z], row k for
        t = start_step-1-k.  Returns the number of model evaluations."""
        B, T, z = x.shape
        assert x.is_contiguous() and x.dtype == torch.float32 and x.device == self.device
        assert table.dtype == torch.float32 and table.is_contiguous() and table.device == self.device and table.shape[1] == 12
        l32 = lengths if (lengths.dtype == torch.int32 and lengths.device == self.device) else _i32(lengths, self.device)
        nz = None
        if noise is not None:
            nz = _f32(noise, self.device)
            assert nz.shape == (start_step, B, T, z), (nz.shape, (start_step, B, T, z))
        self._keep = (l32, table, nz)
        ws = self._workspace(int(self.lib.dn_ddim_workspace_bytes(self.handle, B, T, start_step)))
        wp, wn = self._aligned(ws)
        with torch.cuda.device(self.device):
            return _lib.check(self.lib.dn_ddpm_loop(self.handle, x.data_ptr(), l32.data_ptr(), B, T, start_step, max_evals, table.data_ptr(),
                                                    table.shape[0], int(clip_denoised), int(seed) & (2 ** 64 - 1), _lib.ptr(nz),
                                                    (1 if use_graph else 0) | (2 if split else 0) | (4 if keep_table else 0), wp, wn,
                                                    _lib.current_stream()), "dn_ddpm_loop")


class VaeEngine(_Engine):
    """SpeechVAEEncoderDecoder (reference latent_module.py:1035-1142) on the GPU."""

    def __init__(self, state_dict, dim: int = 768, latent_dim: int = 128, dtype="bf16", device="cuda:0",
                 depth: int = 6, heads: int = 8, dim_head: int = 96, stacks: int = 2, layers: int = 3,
                 vocab: int = 1004):
        self.dim, self.vocab = dim, vocab
        self.mults = packing.vae_mults(latent_dim)
        z = dim
        for m in self.mults:
            z //= m
        self.z = z // 2
        self.dtype = _dtype_code(dtype)
        super().__init__(device, packing.pack_vae(state_dict, dim, self.mults, depth, heads, dim_head, stacks, layers,
                                                  vocab, self.dtype))
        mults = (C.c_int32 * 4)(*(self.mults + [0] * (4 - len(self.mults))))
        c = _lib.VaeConfig(dim, self.z, depth, heads, dim_head, stacks, layers, vocab, len(self.mults), mults, self.dtype)
        _lib.check(self.lib.dn_vae_create(C.byref(c), self._table, len(self.tensors), C.byref(self.handle)),
                   "dn_vae_create")

    def __del__(self):
        if getattr(self, "handle", None) and self.handle.value:
            self.lib.dn_vae_destroy(self.handle)
            self.handle = None

    def workspace_bytes(self, B: int, T: int) -> int:
        return int(self.lib.dn_vae_workspace_bytes(self.handle, B, T))

    def encode_params(self, feat: torch.Tensor) -> torch.Tensor:
        """feat [B,T,dim] fp32 -> posterior parameters [B,T,2z] fp32 ([mean ; logvar])."""
        B, T, _ = feat.shape
        feat = _f32(feat, self.device)
        out = torch.empty(B, T, 2 * self.z, dtype=torch.float32, device=self.device)
        ws = self._workspace(self.workspace_bytes(B, T))
        wp, wn = self._aligned(ws)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dn_vae_encode_params(self.handle, feat.data_ptr(), B, T, out.data_ptr(), wp, wn,
                                                     _lib.current_stream()), "dn_vae_encode_params")
        return out

    def sample_posterior(self, params: torch.Tensor, noise: torch.Tensor, lengths: Optional[torch.Tensor] = None,
                         want_kl: bool = False):
        """DiagonalGaussianDistribution.sample / kl_3d (reference distributions.py:24-41, 62-74)."""
        B, T, _ = params.shape
        noise = _f32(noise, self.device)
        z = torch.empty(B, T, self.z, dtype=torch.float32, device=self.device)
        kl_rows = torch.empty(B, T, dtype=torch.float32, device=self.device) if want_kl else None
        l32 = _i32(lengths, self.device) if lengths is not None else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dn_posterior_sample(params.data_ptr(), 2 * self.z, noise.data_ptr(), self.z, z.data_ptr(),
                                                    None, _lib.DN_F32, self.z, B * T, self.z, T, _lib.ptr(l32),
                                                    _lib.ptr(kl_rows), _lib.current_stream()), "dn_posterior_sample")
        if want_kl:
            return z, kl_rows.sum(dim=1) / (T * self.z)  # mean over (z,T) with pads counted, per sample
        return z

    def encode(self, feat: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
        """encode_feature (reference latent_module.py:1099-1107) with caller-supplied posterior noise."""
        return self.sample_posterior(self.encode_params(feat), noise)

    def decode(self, latent: torch.Tensor, lengths: torch.Tensor, want_recon=True, want_logits=True, want_units=True):
        """decode_feature (:1109-1116) -> (recon [B,T,dim], logits [B,T,vocab], units [B,T] = argmax-4)."""
        B, T, _ = latent.shape
        latent = _f32(latent, self.device)
        l32 = _i32(lengths, self.device)
        recon = torch.empty(B, T, self.dim, dtype=torch.float32, device=self.device) if want_recon else None
        logits = torch.empty(B, T, self.vocab, dtype=torch.float32, device=self.device) if want_logits else None
        units = torch.empty(B, T, dtype=torch.int32, device=self.device) if want_units else None
        ws = self._workspace(self.workspace_bytes(B, T))
        wp, wn = self._aligned(ws)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dn_vae_decode(self.handle, latent.data_ptr(), l32.data_ptr(), B, T, _lib.ptr(recon),
                                              _lib.ptr(logits), _lib.ptr(units), wp, wn, _lib.current_stream()),
                       "dn_vae_decode")
        return recon, logits, units
