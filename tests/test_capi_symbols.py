"""CPU checks of the C-ABI boundary: the library loads, exports every symbol include/diffnorm_hip.h
declares, the ctypes structs match the header's layout, and bad arguments are rejected on the host
(no kernel is launched here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "diffnorm_hip.h")


@pytest.fixture(scope="module")
def lib():
    from diffnorm_amd import _lib

    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    return _lib.load()


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dn_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from diffnorm_amd import _lib

    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
        assert n in _lib.SYMBOLS, f"{n} declared in the header but not bound in _lib.SYMBOLS"
    assert sorted(_lib.SYMBOLS) == names


def test_struct_layout_matches_header(lib, tmp_path):
    """Compiles the header with gcc and compares sizeof/offsetof with the ctypes mirrors."""
    import subprocess

    from diffnorm_amd import _lib

    probes = {
        "DnGemmTerm": (_lib.GemmTerm, ["A", "W", "lda", "shift", "a_gstride", "w_gstride", "shift_by_group", "layout"]),
        "DnGemmParams": (_lib.GemmParams, ["terms", "n_terms", "dtype", "M", "N", "K", "T", "groups", "epilogue", "bias",
                                           "bias_gstride", "out", "ldo", "out_dtype", "out_gstride", "res", "ldr", "res_dtype",
                                           "res_gstride", "gamma_beta", "gb_ld", "gb_half", "gb_gstride", "pos_table", "pos_ld",
                                           "lengths", "norm_out", "norm_ld", "norm_dtype", "norm_D", "norm_gb_ld", "norm_gamma",
                                           "norm_gb", "norm_gb_half", "out_layout", "norm_split", "norm_ssq_ld", "norm_ssq", "row_ssq",
                                           "row_ssq_ld", "row_ssq_parts", "row_D", "row_bias_ld", "row_bias"]),
        "DnAdamParams": (_lib.AdamParams, ["lr", "beta1", "beta2", "eps", "weight_decay", "max_norm", "step", "grad_scale",
                                           "grad_scale_dev"]),
        "DnAttnParams": (_lib.AttnParams, ["q", "k", "v", "out", "ldq", "ldk", "ldv", "ldo", "B", "T", "heads", "dim_head",
                                           "dtype", "Tk", "lengths", "scale", "lse", "dropout_p", "seed_lo", "seed_hi"]),
        "DnAttnBwdParams": (_lib.AttnBwdParams, ["q", "k", "v", "out", "dout", "dq", "dk", "dv", "ldq", "ldk", "ldv", "ldo", "lddo", "lddq",
                                                 "lddk", "lddv", "B", "T", "heads", "dim_head", "dtype", "lengths", "scale", "lse",
                                                 "delta", "dropout_p", "seed_lo", "seed_hi"]),
        "DnGaussianMoments": (_lib.GaussianMoments, ["x", "model_out", "x_start", "t", "table", "mean", "variance", "log_variance",
                                                      "pred_xstart", "vb", "reverse_sample", "N", "inner", "learned_range", "clip_denoised"]),
        "DnVaeTrainBatch": (_lib.VaeTrainBatch, ["feat", "units", "lengths", "noise", "B", "T", "ntokens", "w_lsce", "w_mse", "w_kl",
                                                 "label_smoothing", "loss_scale", "stats", "logits_out", "recon_out", "ext_dlogits"]),
        "DnEpsTrainBatch": (_lib.EpsTrainBatch, ["feat", "units", "lengths", "z", "jitter", "true_noise", "times", "sqrt_ac", "sqrt_1mac",
                                                 "snr_weight", "beta0", "B", "T", "n_units", "n_frames", "timesteps", "multitask",
                                                 "label_smoothing", "recon_weight", "loss_scale", "stats", "eps_out"]),
        "DnEpsConfig": (_lib.EpsConfig, ["dim", "latent", "depth", "heads", "dim_head", "wn_layers", "wn_stacks", "cond_mult",
                                         "dtype", "max_pos", "dim_prompt", "num_latents", "resampler_depth"]),
        "DnVaeConfig": (_lib.VaeConfig, ["dim", "z", "depth", "heads", "dim_head", "stacks", "layers", "vocab", "n_mults",
                                         "mults", "dtype"]),
    }
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for cname, (_, fields) in probes.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for f in fields:
            lines.append(f'printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    lines.append("return 0;}")
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-o", str(exe), str(src)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, (ct, fields) in probes.items():
        assert C.sizeof(ct) == int(out[cname]), cname
        for f in fields:
            assert getattr(ct, f).offset == int(out[f"{cname}.{f}"]), f"{cname}.{f}"


def test_version_and_host_side_argument_checks(lib):
    from diffnorm_amd import _lib

    assert lib.dn_version() >= 100
    p = _lib.GemmParams()  # all zeros: must be refused before any launch
    assert lib.dn_conv_gemm(C.byref(p), None) == -1
    assert b"dn_conv_gemm" in lib.dn_last_error()
    a = _lib.AttnParams()
    assert lib.dn_attention(C.byref(a), None) == -1
    assert lib.dn_eps_workspace_bytes(None, 1, 1) == 0
    with pytest.raises(_lib.DiffNormHipError):
        _lib.check(-1, "probe")


def test_engines_refuse_to_run_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from diffnorm_amd import _lib, engine

    with pytest.raises(_lib.DiffNormHipError):
        engine._require_cuda("cuda:0")
    with pytest.raises(_lib.DiffNormHipError):
        engine._require_cuda("cpu")


def test_no_kernel_spills_or_uses_scratch():
    """Resource check of the compiled kernels (no GPU needed: hipcc cross-compiles) of the contraction units (one per arithmetic) and
    attention.hip: no scratch access inside any loop and no stack object -- a spill inside a hand-scheduled K loop is both a
    slowdown and a hazard (scratch loads share vmcnt with the counted DMA waits).  Registers parked around a kernel's loops (the
    256-register split-operand tile carries a dozen epilogue constants that way) are reported by the tool, not failed."""
    import importlib.util
    import shutil

    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    spec = importlib.util.spec_from_file_location("check_resources", os.path.join(ROOT, "tools", "check_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.check(["gemm_bf16.hip", "gemm_f32.hip", "gemm_x3.hip", "attention.hip", "wgrad_tn.hip"]) == []


def test_tile_choice_is_host_logic(lib, monkeypatch):
    """dn_conv_gemm_tile / dn_conv_gemm_kblocked_ok read only the shape fields (no GPU): the routing the engines rely on when
    they decide which buffers to lay out K-blocked -- the shapes of one denoising step at [32,512] and their small-batch cases."""
    from diffnorm_amd import _lib

    for v in ("DN_GEMM_TILE", "DN_GEMM_HEUR"):
        monkeypatch.delenv(v, raising=False)

    def params(M, N, K, T=512, groups=1, n_terms=1, epi=_lib.EPI_BIAS, dtype=_lib.DN_BF16, tile=0, layout=0):
        p = _lib.GemmParams()
        p.M, p.N, p.K, p.T, p.groups, p.n_terms, p.epilogue, p.dtype = M, N, K, T, groups, n_terms, epi, dtype
        p.pad_ = tile << 16
        for i in range(n_terms):
            p.terms[i].layout = layout
        return p

    tile = lambda p: lib.dn_conv_gemm_tile(C.byref(p))
    M = 32 * 512
    ffn_conv = params(M, 1408, 1408, n_terms=3)
    assert tile(ffn_conv) == 4 and lib.dn_conv_gemm_kblocked_ok(C.byref(ffn_conv)) == 1      # 256 x 352: 1408 = 4 x 352
    assert tile(params(M, 1408, 1408, n_terms=3, dtype=_lib.DN_F32)) != 4                      # bf16 only
    assert tile(params(2 * 100, 1408, 1408, T=100, n_terms=3)) != 4                            # too few tiles to own whole CUs
    assert lib.dn_conv_gemm_kblocked_ok(C.byref(params(2 * 100, 1408, 1408, T=100, n_terms=3))) == 0
    assert tile(params(M, 512, 512, groups=8)) == 3                                            # WaveNet res conv: 256 x 256
    assert tile(params(M, 512, 512, groups=8, n_terms=3, epi=_lib.EPI_FILM_GATE)) == 3         # dilated conv + FiLM gate
    assert tile(params(M, 1408, 512, epi=_lib.EPI_GEGLU)) == 3                                 # GEGLU projection (2816 packed columns)
    assert tile(params(M, 1536, 512)) == 1                                                     # q/kv: two 128 x 128 workgroups per CU
    assert tile(params(M, 2048, 2048, n_terms=3)) == 3                                         # the VAE's FFN conv (not a multiple of 352)
    # a lone launch whose 256 x 256 tiles leave most of a round idle goes to the 256 x 192 form when the width divides (training)
    assert tile(params(24 * 512, 768, 768)) == 8 and tile(params(24 * 512, 768, 2048, epi=_lib.EPI_RESADD)) == 8
    _lib.set_option("tile_192", 0)
    try:
        assert tile(params(24 * 512, 768, 768)) == 3
    finally:
        _lib.set_option("tile_192", None)
    # K-blocked operands: taken by the two 256-row tiles; a forced tile that cannot take them is an error (-1)
    assert tile(params(M, 1536, 512, layout=3)) == 3
    assert tile(params(M, 1408, 1408, n_terms=3, layout=3)) == 4
    assert tile(params(M, 1536, 512, layout=3, tile=1)) == -1
    assert tile(params(M, 1536, 512, layout=1, dtype=_lib.DN_F32)) == -1
    for forced in (1, 2, 3):
        assert tile(params(M, 1536, 512, tile=forced)) == forced
