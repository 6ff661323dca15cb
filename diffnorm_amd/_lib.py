"""ctypes binding of libdiffnorm_hip.so (C ABI: include/diffnorm_hip.h).

The product path has no CPU fallback: if the library is missing or a call fails, this module
raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C diffnorm_amd/csrc``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdiffnorm_hip.so")

DN_F32, DN_BF16, DN_BF16X3, DN_F16 = 0, 1, 2, 3
EPI_BIAS, EPI_SILU, EPI_GEGLU, EPI_FILM_GATE, EPI_RESADD, EPI_POSEMB, EPI_RELU = range(7)
DN_MAX_TERMS = 8
TAG_FFN_CONV, TAG_WN_DILATED, TAG_FFN_CONV_WGRAD = 1, 2, 3


class DiffNormHipError(RuntimeError):
    pass


LAYOUT_A_KBLOCKED, LAYOUT_W_KBLOCKED, LAYOUT_OUT_KBLOCKED = 1, 2, 1


class GemmTerm(C.Structure):
    _fields_ = [("A", C.c_void_p), ("W", C.c_void_p), ("lda", C.c_int32), ("shift", C.c_int32),
                ("a_gstride", C.c_int64), ("w_gstride", C.c_int64), ("shift_by_group", C.c_int32),
                ("layout", C.c_int32), ("ldw", C.c_int32), ("pad_ldw_", C.c_int32)]


class GemmParams(C.Structure):
    _fields_ = [("terms", GemmTerm * DN_MAX_TERMS), ("n_terms", C.c_int32), ("dtype", C.c_int32),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("T", C.c_int32),
                ("groups", C.c_int32), ("epilogue", C.c_int32), ("bias", C.c_void_p),
                ("bias_gstride", C.c_int64), ("out", C.c_void_p), ("ldo", C.c_int32),
                ("out_dtype", C.c_int32), ("out_gstride", C.c_int64), ("res", C.c_void_p),
                ("ldr", C.c_int32), ("res_dtype", C.c_int32), ("res_gstride", C.c_int64),
                ("gamma_beta", C.c_void_p), ("gb_ld", C.c_int32), ("gb_half", C.c_int32),
                ("gb_gstride", C.c_int64), ("pos_table", C.c_void_p), ("pos_ld", C.c_int32),
                ("pad_", C.c_int32), ("lengths", C.c_void_p),
                ("norm_out", C.c_void_p), ("norm_ld", C.c_int32), ("norm_dtype", C.c_int32), ("norm_D", C.c_int32),
                ("norm_gb_ld", C.c_int32), ("norm_gamma", C.c_void_p), ("norm_gb", C.c_void_p),
                ("norm_gb_half", C.c_int32), ("out_layout", C.c_int32),
                ("norm_split", C.c_int32), ("norm_ssq_ld", C.c_int32), ("norm_ssq", C.c_void_p),
                ("row_ssq", C.c_void_p), ("row_ssq_ld", C.c_int32), ("row_ssq_parts", C.c_int32),
                ("row_D", C.c_float), ("row_bias_ld", C.c_int32), ("row_bias", C.c_void_p),
                ("pre_out", C.c_void_p), ("pre_ld", C.c_int32), ("pre_pad_", C.c_int32)]


class AdamParams(C.Structure):
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("weight_decay", C.c_double), ("max_norm", C.c_double), ("step", C.c_int32), ("pad_", C.c_int32),
                ("grad_scale", C.c_double), ("grad_scale_dev", C.c_void_p)]


class AttnParams(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("out", C.c_void_p),
                ("ldq", C.c_int32), ("ldk", C.c_int32), ("ldv", C.c_int32), ("ldo", C.c_int32),
                ("B", C.c_int32), ("T", C.c_int32), ("heads", C.c_int32), ("dim_head", C.c_int32),
                ("dtype", C.c_int32), ("Tk", C.c_int32), ("lengths", C.c_void_p),
                ("scale", C.c_float), ("pad2_", C.c_int32), ("lse", C.c_void_p),
                ("dropout_p", C.c_float), ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32), ("pad3_", C.c_int32)]


class AttnBwdParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("q", "k", "v", "out", "dout", "dq", "dk", "dv")] + \
               [(n, C.c_int32) for n in ("ldq", "ldk", "ldv", "ldo", "lddo", "lddq", "lddk", "lddv", "B", "T", "heads", "dim_head",
                                         "dtype", "pad_")] + \
               [("lengths", C.c_void_p), ("scale", C.c_float), ("pad2_", C.c_int32), ("lse", C.c_void_p), ("delta", C.c_void_p),
                ("dropout_p", C.c_float), ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32), ("pad3_", C.c_int32)]


class GaussianStep(C.Structure):
    _fields_ = [("x", C.c_void_p), ("model_out", C.c_void_p), ("noise", C.c_void_p), ("sample", C.c_void_p),
                ("pred_xstart", C.c_void_p), ("t", C.c_void_p), ("table", C.c_void_p), ("N", C.c_int32),
                ("inner", C.c_int32), ("learned_range", C.c_int32), ("clip_denoised", C.c_int32), ("sampler", C.c_int32),
                ("eta", C.c_float), ("predict_xstart", C.c_int32), ("cond_grad", C.c_void_p)]


GD_COLS = 12


class GaussianMoments(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x", "model_out", "x_start", "t", "table", "mean", "variance", "log_variance", "pred_xstart",
                                          "vb", "reverse_sample")] + [(n, C.c_int32) for n in ("N", "inner", "learned_range", "clip_denoised", "predict_xstart")]


class EpsConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dim", "latent", "depth", "heads", "dim_head", "wn_layers",
                                         "wn_stacks", "cond_mult", "dtype", "max_pos", "dim_prompt", "num_latents", "resampler_depth")]


class VaeConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dim", "z", "depth", "heads", "dim_head", "stacks", "layers",
                                         "vocab", "n_mults")] + [("mults", C.c_int32 * 4), ("dtype", C.c_int32)]


class NarConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dim", "ffn", "layers", "heads", "vocab", "max_pos", "pad", "dtype")]


class NarEncConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("input_dim", "conv_channels", "kernel", "dim", "ffn", "layers", "heads", "max_pos", "pad", "dtype")]


class VaeTrainBatch(C.Structure):
    _fields_ = [("feat", C.c_void_p), ("units", C.c_void_p), ("lengths", C.c_void_p), ("noise", C.c_void_p), ("B", C.c_int32),
                ("T", C.c_int32), ("ntokens", C.c_int32), ("w_lsce", C.c_float), ("w_mse", C.c_float), ("w_kl", C.c_float),
                ("label_smoothing", C.c_float), ("loss_scale", C.c_float), ("stats", C.c_void_p), ("logits_out", C.c_void_p),
                ("recon_out", C.c_void_p), ("ext_dlogits", C.c_void_p), ("attn_dropout", C.c_float),
                ("dropout_seed_lo", C.c_uint32), ("dropout_seed_hi", C.c_uint32), ("pad_", C.c_int32)]


class EpsTrainBatch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("feat", "units", "lengths", "z", "jitter", "true_noise", "times", "sqrt_ac", "sqrt_1mac",
                                          "snr_weight")] + \
               [("beta0", C.c_float), ("B", C.c_int32), ("T", C.c_int32), ("n_units", C.c_int32), ("n_frames", C.c_int32),
                ("timesteps", C.c_int32), ("multitask", C.c_int32), ("label_smoothing", C.c_float), ("recon_weight", C.c_float),
                ("loss_scale", C.c_float), ("stats", C.c_void_p), ("eps_out", C.c_void_p), ("attn_dropout", C.c_float),
                ("dropout_seed_lo", C.c_uint32), ("dropout_seed_hi", C.c_uint32), ("pad_", C.c_int32)]


# every symbol include/diffnorm_hip.h declares: name -> (restype, argtypes)
_vp, _i32, _i64, _u64, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_size_t
SYMBOLS = {
    "dn_conv_gemm": (C.c_int, [C.POINTER(GemmParams), _vp]),
    "dn_conv_gemm_kblocked_ok": (C.c_int, [C.POINTER(GemmParams)]),
    "dn_conv_gemm_tile": (C.c_int, [C.POINTER(GemmParams)]),
    "dn_grad_sumsq": (C.c_int, [_vp, C.c_int64, _vp, _vp, C.c_int32, _vp]),
    "dn_transpose_pad": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, C.c_int32, C.c_int32,
                                   C.c_int32, C.c_int32, _vp]),
    "dn_adam_step": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.POINTER(AdamParams), _vp, _vp, _vp]),
    "dn_profile_start": (C.c_int, [_i32, _i32]),
    "dn_profile_stop": (C.c_int, [C.POINTER(C.c_float), C.POINTER(_i32)]),
    "dn_attention": (C.c_int, [C.POINTER(AttnParams), _vp]),
    "dn_attention_backward": (C.c_int, [C.POINTER(AttnBwdParams), _vp]),
    "dn_time_cond_backward": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "dn_gate_forward": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp]),
    "dn_gate_backward": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _vp]),
    "dn_geglu_forward": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "dn_geglu_backward": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "dn_rmsnorm_backward_scratch_bytes": (_sz, [_i32, _i32, _i32]),
    "dn_rmsnorm_backward": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32,
                                      _vp, _vp, _i32, _vp, _vp]),
    "dn_colsum_scratch_bytes": (_sz, [_i32, _i32, _i32]),
    "dn_colsum": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, C.c_float, _i32, _vp, _vp]),
    "dn_posterior_backward": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, C.c_float, _vp]),
    "dn_lsce_loss_grad": (C.c_int, [_vp, _i32, _vp, _i32, _i32, C.c_float, C.c_float, _vp, _vp, _i32, _i32, _vp]),
    "dn_masked_mse_grad": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, C.c_float, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _vp]),
    "dn_vec_sum": (C.c_int, [_vp, _i64, _vp, _i32, _vp, _vp]),
    "dn_vae_train_create": (C.c_int, [C.POINTER(VaeConfig), C.POINTER(_vp)]),
    "dn_vae_train_destroy": (None, [_vp]),
    "dn_vae_train_param_count": (_i64, [_vp]),
    "dn_vae_train_aux_bytes": (_sz, [_vp]),
    "dn_vae_train_offsets": (C.c_int, [_vp, C.POINTER(_i64), _i32]),
    "dn_vae_train_stage_range": (C.c_int, [_vp, _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    "dn_vae_train_bind": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "dn_vae_train_refresh": (C.c_int, [_vp, _vp]),
    "dn_vae_train_workspace_bytes": (_sz, [_vp, _i32, _i32]),
    "dn_vae_train_forward": (C.c_int, [_vp, C.POINTER(VaeTrainBatch), _vp, _sz, _vp]),
    "dn_vae_train_backward": (C.c_int, [_vp, C.POINTER(VaeTrainBatch), _i32, _i32, _vp, _sz, _vp]),
    "dn_cmlm_step": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dn_eps_train_create": (C.c_int, [C.POINTER(EpsConfig), C.POINTER(_vp)]),
    "dn_eps_train_destroy": (None, [_vp]),
    "dn_eps_train_param_count": (_i64, [_vp]),
    "dn_eps_train_aux_bytes": (_sz, [_vp]),
    "dn_eps_train_offsets": (C.c_int, [_vp, C.POINTER(_i64), _i32]),
    "dn_eps_train_stage_range": (C.c_int, [_vp, _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    "dn_eps_train_bind": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "dn_eps_train_refresh": (C.c_int, [_vp, _vp]),
    "dn_eps_train_workspace_bytes": (_sz, [_vp, _vp, _i32, _i32]),
    "dn_eps_train_forward": (C.c_int, [_vp, _vp, C.POINTER(EpsTrainBatch), _vp, _sz, _vp]),
    "dn_eps_train_backward": (C.c_int, [_vp, _vp, C.POINTER(EpsTrainBatch), _i32, _i32, _vp, _sz, _vp]),
    "dn_sum_groups": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _i64, _vp]),
    "dn_rows_times_weight_scratch_bytes": (_sz, [_i32, _i32, _i32]),
    "dn_rows_times_weight": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "dn_transpose_weights": (C.c_int, [_vp, _i32, _i32, _i64, _i32, _i32, _vp, _i64, _i32, _i32, _vp]),
    "dn_wgrad_reduce": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp]),
    "dn_conv_weight_grad_tn": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "dn_add_broadcast": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "dn_transpose_pad_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp]),
    "dn_rmsnorm": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _vp]),
    "dn_time_cond": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _vp]),
    "dn_ddim_step": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "dn_q_sample": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "dn_gaussian_step": (C.c_int, [C.POINTER(GaussianStep), _vp]),
    "dn_gaussian_moments": (C.c_int, [C.POINTER(GaussianMoments), _vp]),
    "dn_posterior_sample": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "dn_argmax_units": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "dn_randn": (C.c_int, [_vp, _i64, _u64, _u64, _vp]),
    "dn_convert_rows": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp]),
    "dn_eps_create": (C.c_int, [C.POINTER(EpsConfig), C.POINTER(_vp), _i32, C.POINTER(_vp)]),
    "dn_eps_destroy": (None, [_vp]),
    "dn_eps_workspace_bytes": (_sz, [_vp, _i32, _i32]),
    "dn_eps_forward": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "dn_eps_cond_workspace_bytes": (_sz, [_vp, _i32, _i32, _i32]),
    "dn_eps_forward_cond": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "dn_eps_forward_cond_ex": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _sz, _i32, _vp, _i32, _i32, _vp]),
    "dn_eps_cond_time_table_workspace_bytes": (_sz, [_vp, _i32]),
    "dn_eps_cond_time_table": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _sz, _vp]),
    "dn_vae_create": (C.c_int, [C.POINTER(VaeConfig), C.POINTER(_vp), _i32, C.POINTER(_vp)]),
    "dn_vae_destroy": (None, [_vp]),
    "dn_vae_workspace_bytes": (_sz, [_vp, _i32, _i32]),
    "dn_vae_encode_params": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _sz, _vp]),
    "dn_vae_decode": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dn_ddim_workspace_bytes": (_sz, [_vp, _i32, _i32, _i32]),
    "dn_ddim_loop": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _sz, _vp]),
    "dn_ddpm_loop": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, C.c_uint64, _vp, _i32, _vp, _sz, _vp]),
    "dn_cfg_combine": (C.c_int, [_vp, C.c_float, _i64, _vp, _vp]),
    "dn_nar_create": (C.c_int, [C.POINTER(NarConfig), C.POINTER(_vp), _i32, C.POINTER(_vp)]),
    "dn_nar_destroy": (None, [_vp]),
    "dn_nar_workspace_bytes": (_sz, [_vp, _i32, _i32, _i32]),
    "dn_nar_cross_kv_bytes": (_sz, [_vp, _i32, _i32]),
    "dn_nar_cross_kv": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _sz, _vp]),
    "dn_nar_predict_lengths": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _sz, _vp]),
    "dn_nar_decoder_forward": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "dn_nar_encoder_create": (C.c_int, [_vp, _vp, _i32, _vp]),
    "dn_nar_encoder_destroy": (None, [_vp]),
    "dn_nar_encoder_out_frames": (_i32, [_i32]),
    "dn_nar_encoder_workspace_bytes": (_sz, [_vp, _i32, _i32]),
    "dn_nar_encoder_forward": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "dn_last_error": (C.c_char_p, []),
    "dn_version": (C.c_int, []),
    "dn_cmlm_step_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _vp]),
    "dn_set_option": (C.c_int, [C.c_char_p, _i32]),
    "dn_get_option": (C.c_int, [C.c_char_p, _vp, _vp]),
}

OPTION_DEFAULT = -2147483648


def set_option(name: str, value) -> None:
    """dn_set_option: a process-wide run-time option of the library (include/diffnorm_hip.h); value None = back to its start value."""
    check(load().dn_set_option(name.encode(), OPTION_DEFAULT if value is None else int(value)), "dn_set_option")


def get_option(name: str):
    """Current value of a run-time option, or None when it is neither set nor in the environment."""
    v, on = C.c_int32(), C.c_int32()
    check(load().dn_get_option(name.encode(), C.addressof(v), C.addressof(on)), "dn_get_option")
    return v.value if on.value else None


class option:
    """with _lib.option("taps_inner", 2): ...  -- sets a run-time option and restores what was there before."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.prev = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.prev)
        return False

_lib = None


def load():
    """Loads the shared library (once) and binds every declared symbol.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise DiffNormHipError(
            f"{LIB_PATH} not found: the HIP library is required (no CPU fallback). "
            "Build it with `make -C diffnorm_amd/csrc` or `__graft_entry__.build()`.")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and every device pointer this library receives
    # comes from torch, so torch's copy must be the one that is loaded (and initialised) first -- loading this library
    # first binds it to the system copy, which then has no device context ("no ROCm-capable device is detected").
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc < 0:
        msg = load().dn_last_error()
        raise DiffNormHipError(f"{what or 'libdiffnorm_hip'} failed ({rc}): {msg.decode() if msg else ''}")
    return rc


def ptr(t):
    """Device pointer of a torch tensor (or 0 for None)."""
    return 0 if t is None else t.data_ptr()


def current_stream():
    import torch

    return torch.cuda.current_stream().cuda_stream
