// dn_conv_gemm: argument checks and the C entry points; the kernels live in gemm_kernels.h and are instantiated per arithmetic
// in gemm_bf16.hip / gemm_f32.hip (one translation unit each, so they build in parallel).
#include "gemm_kernels.h"

namespace dn {
LaunchProfile g_prof;
thread_local bool g_gemm_twin = false;
}

extern "C" int dn_conv_gemm_kblocked_ok(const DnGemmParams* pp) {
  return pp && pp->K % 32 == 0 && dn::routes_to_352(*pp) ? 1 : 0;
}

extern "C" int dn_conv_gemm_tile(const DnGemmParams* pp) { return pp ? dn::choose_tile(*pp) : -1; }

extern "C" int dn_conv_gemm(const DnGemmParams* pp, void* stream) {
  DN_CHECK_ARG(pp != nullptr, "dn_conv_gemm: null params");
  const DnGemmParams& p = *pp;
  DN_CHECK_ARG(p.dtype == DN_F32 || p.dtype == DN_BF16 || p.dtype == DN_BF16X3 || p.dtype == DN_F16, "dn_conv_gemm: bad dtype %d", p.dtype);
  const bool h16 = dn::dn_is16(p.dtype);
  const int kt = h16 ? 64 : 32;
  const bool x3 = p.dtype == DN_BF16X3;
  DN_CHECK_ARG(p.n_terms >= 1 && p.n_terms <= DN_MAX_TERMS, "dn_conv_gemm: n_terms %d", p.n_terms);
  DN_CHECK_ARG(p.M > 0 && p.N > 0 && p.T > 0 && p.groups >= 1, "dn_conv_gemm: M=%d N=%d T=%d groups=%d", p.M, p.N, p.T, p.groups);
  DN_CHECK_ARG(p.M % p.T == 0, "dn_conv_gemm: M=%d is not a multiple of T=%d", p.M, p.T);
  DN_CHECK_ARG(p.K > 0 && p.K % kt == 0, "dn_conv_gemm: K=%d must be a multiple of %d", p.K, kt);
  DN_CHECK_ARG(p.N % 4 == 0 && p.ldo % 4 == 0 && p.ldo >= p.N, "dn_conv_gemm: N=%d ldo=%d must be multiples of 4, ldo>=N", p.N, p.ldo);
  DN_CHECK_ARG(p.out != nullptr, "dn_conv_gemm: null out");
  for (int i = 0; i < p.n_terms; ++i) {
    DN_CHECK_ARG(p.terms[i].A && p.terms[i].W, "dn_conv_gemm: term %d null operand", i);
    DN_CHECK_ARG(p.terms[i].lda >= p.K && p.terms[i].lda % (16 / (h16 ? 2 : 4)) == 0,
                 "dn_conv_gemm: term %d lda=%d (K=%d)", i, p.terms[i].lda, p.K);
    DN_CHECK_ARG(p.terms[i].shift > -p.T && p.terms[i].shift < p.T + (1 << 20), "dn_conv_gemm: term %d shift %d", i, p.terms[i].shift);
    DN_CHECK_ARG((reinterpret_cast<uintptr_t>(p.terms[i].A) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.terms[i].W) & 15) == 0,
                 "dn_conv_gemm: term %d operands must be 16-byte aligned", i);
  }
  if (x3) {
    DN_CHECK_ARG(!(p.norm_out && !p.norm_split), "dn_conv_gemm: the whole-row fused norm is not built for split operands (use norm_split)");
    for (int i = 0; i < p.n_terms; ++i)
      DN_CHECK_ARG(p.terms[i].lda % 32 == 0 && p.terms[i].a_gstride % 32 == 0 && p.terms[i].w_gstride % 32 == 0,
                   "dn_conv_gemm: term %d: split rows need lda / group strides that are multiples of 32 elements", i);
    DN_CHECK_ARG(p.epilogue != DN_EPI_FILM_GATE || p.res_dtype == DN_F32, "dn_conv_gemm: split operands: the FiLM epilogue's residual input is fp32");
  }
  // a kernel writes fp32 or its own operand type: bf16 from DN_BF16, half from DN_F16, split rows from DN_BF16X3
  DN_CHECK_ARG((p.out_dtype == DN_F16) == (p.dtype == DN_F16 && p.out_dtype != DN_F32) && (p.dtype != DN_F16 || p.out_dtype == DN_F32 || p.out_dtype == DN_F16),
               "dn_conv_gemm: out_dtype %d from dtype %d (half tensors come from and go to DN_F16 contractions only)", p.out_dtype, p.dtype);
  if (p.epilogue == DN_EPI_FILM_GATE)
    DN_CHECK_ARG(p.dtype == DN_F16 ? (p.res_dtype == DN_F32 || p.res_dtype == DN_F16) : p.res_dtype != DN_F16, "dn_conv_gemm: res_dtype %d with dtype %d", p.res_dtype, p.dtype);
  if (p.out_dtype == DN_BF16X3)
    DN_CHECK_ARG(p.ldo % 32 == 0 && p.out_gstride % 32 == 0 && ((uintptr_t)p.out & 127) == 0 && !p.out_layout,
                 "dn_conv_gemm: a split-row output needs ldo / out_gstride multiples of 32 elements and a 128-byte aligned base");
  if (p.norm_out && p.norm_dtype == DN_BF16X3)
    DN_CHECK_ARG(p.norm_ld % 32 == 0 && ((uintptr_t)p.norm_out & 127) == 0 && p.norm_split != 2,
                 "dn_conv_gemm: a split-row norm_out needs norm_ld a multiple of 32 and a 128-byte aligned base");
  if (p.epilogue == DN_EPI_FILM_GATE || p.epilogue == DN_EPI_RESADD)
    DN_CHECK_ARG(p.res != nullptr && p.ldr % 4 == 0, "dn_conv_gemm: epilogue needs res (ldr multiple of 4)");
  if (p.epilogue == DN_EPI_RESADD) DN_CHECK_ARG(p.out_dtype == DN_F32, "dn_conv_gemm: RESADD writes fp32");
  if (p.epilogue == DN_EPI_POSEMB) DN_CHECK_ARG(p.pos_table && p.lengths && p.pos_ld % 4 == 0, "dn_conv_gemm: POSEMB needs pos_table and lengths");
  if (p.epilogue == DN_EPI_FILM_GATE && p.gamma_beta) DN_CHECK_ARG(p.gb_half % 4 == 0 && p.gb_ld % 4 == 0, "dn_conv_gemm: gamma_beta strides must be multiples of 4");
  if (p.norm_split) {
    DN_CHECK_ARG(p.epilogue == DN_EPI_RESADD || p.epilogue == DN_EPI_POSEMB, "dn_conv_gemm: norm_split needs a RESADD or POSEMB epilogue");
    DN_CHECK_ARG(p.norm_split == 1 || (p.norm_split == 2 && h16 && p.norm_dtype == p.dtype && p.norm_ld % 32 == 0),
                 "dn_conv_gemm: norm_split=%d (2 = K-blocked bf16 norm_out, norm_ld a multiple of 32)", p.norm_split);
    DN_CHECK_ARG(p.norm_out && p.norm_ssq && p.N % 64 == 0 && p.norm_ld % 4 == 0 && p.norm_ld >= p.N && p.norm_ssq_ld * 64 >= p.N,
                 "dn_conv_gemm: norm_split needs norm_out, norm_ssq and N a multiple of 64 (N=%d)", p.N);
    DN_CHECK_ARG((p.norm_dtype == DN_F32 || p.norm_dtype == DN_BF16 || p.norm_dtype == DN_BF16X3 || p.norm_dtype == DN_F16) && (p.norm_dtype == DN_F16) == (p.dtype == DN_F16 && p.norm_dtype != DN_F32),
                 "dn_conv_gemm: norm_dtype %d with dtype %d", p.norm_dtype, p.dtype);
    DN_CHECK_ARG(!p.norm_gb || p.norm_gb_ld % 4 == 0, "dn_conv_gemm: norm_gb stride must be a multiple of 4");
  } else if (p.norm_out) {
    DN_CHECK_ARG(p.epilogue == DN_EPI_RESADD || p.epilogue == DN_EPI_POSEMB, "dn_conv_gemm: norm_out needs a RESADD or POSEMB epilogue");
    DN_CHECK_ARG(p.N <= 512 && p.out_dtype == DN_F32, "dn_conv_gemm: the fused norm needs N <= 512 and an fp32 stream");
    DN_CHECK_ARG(p.norm_D > 0 && p.norm_D <= p.N && p.norm_D % 4 == 0 && p.norm_ld % 4 == 0 && p.norm_ld >= p.norm_D && p.norm_ld <= 512,
                 "dn_conv_gemm: bad norm_D=%d / norm_ld=%d", p.norm_D, p.norm_ld);
    DN_CHECK_ARG(!p.norm_gb || (p.norm_gb_ld % 4 == 0 && p.norm_gb_half % 4 == 0), "dn_conv_gemm: norm_gb strides must be multiples of 4");
  }
  for (int i = 0; i < p.n_terms; ++i)
    DN_CHECK_ARG(p.terms[i].ldw == 0 || (p.terms[i].ldw >= p.K && p.terms[i].layout == 0 && p.dtype != DN_BF16X3 && !(p.norm_out && !p.norm_split)),
                 "dn_conv_gemm: term %d: ldw=%d needs ldw >= K, row-major operands, bf16 / f32 and no whole-row fused norm", i, p.terms[i].ldw);
  for (int i = 0; i < p.n_terms; ++i)
    DN_CHECK_ARG((p.terms[i].layout & ~3) == 0 && (p.terms[i].layout == 0 || h16),
                 "dn_conv_gemm: term %d layout=%d (K-blocked operands: 2-byte types only)", i, p.terms[i].layout);
  if (p.out_layout)
    DN_CHECK_ARG(p.out_layout == DN_LAYOUT_OUT_KBLOCKED && h16 && p.out_dtype == p.dtype && p.N % 32 == 0 && !p.norm_out &&
                     (p.epilogue == DN_EPI_BIAS || p.epilogue == DN_EPI_SILU || p.epilogue == DN_EPI_RELU || p.epilogue == DN_EPI_GEGLU || p.epilogue == DN_EPI_FILM_GATE),
                 "dn_conv_gemm: K-blocked output needs a BIAS, SILU, GEGLU or FILM_GATE epilogue, a bf16 destination and N a multiple of 32");
  if (p.row_ssq) {
    DN_CHECK_ARG(p.epilogue == DN_EPI_BIAS || p.epilogue == DN_EPI_SILU || p.epilogue == DN_EPI_RELU || p.epilogue == DN_EPI_GEGLU,
                 "dn_conv_gemm: row_ssq (split norm) needs a BIAS, SILU or GEGLU epilogue");
    DN_CHECK_ARG(p.row_ssq_parts >= 1 && p.row_ssq_ld >= p.row_ssq_parts && p.row_D > 0.f, "dn_conv_gemm: bad row_ssq_parts / row_ssq_ld / row_D");
    DN_CHECK_ARG(p.row_bias_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(p.row_bias) & 15) == 0, "dn_conv_gemm: row_bias must be 16-byte aligned, stride a multiple of 4");
  }
  if (p.pre_out)
    DN_CHECK_ARG(p.epilogue == DN_EPI_GEGLU && p.groups == 1 && p.dtype != DN_BF16X3 && p.out_dtype == p.dtype && p.pre_ld >= 2 * p.N && p.pre_ld % 8 == 0 &&
                     (reinterpret_cast<uintptr_t>(p.pre_out) & 15) == 0,
                 "dn_conv_gemm: pre_out (the kept pre-activation) needs the GEGLU epilogue, one group, out_dtype == dtype, pre_ld >= 2 N and a multiple of 8");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (p.dtype == DN_BF16X3) return dn::gemm_dispatch_x3(p, s);
  if (p.dtype == DN_F16) return dn::gemm_dispatch_f16(p, s);
  return p.dtype == DN_BF16 ? dn::gemm_dispatch_bf16(p, s) : dn::gemm_dispatch_f32(p, s);
}

extern "C" int dn_profile_start(int32_t tag, int32_t max_launches) {
  DN_CHECK_ARG(tag > 0 && tag < 256 && max_launches > 0 && max_launches <= 4096, "dn_profile_start: bad arguments");
  DN_CHECK_ARG(dn::g_prof.cap == 0, "dn_profile_start: a profile is already open");
  dn::g_prof.ev = new hipEvent_t[2 * max_launches];
  for (int i = 0; i < 2 * max_launches; ++i)
    if (hipEventCreate(&dn::g_prof.ev[i]) != hipSuccess) {
      dn_set_error("dn_profile_start: hipEventCreate failed");
      return DN_ELAUNCH;
    }
  dn::g_prof.tag = tag; dn::g_prof.n = 0; dn::g_prof.cap = max_launches;
  return DN_OK;
}

extern "C" int dn_profile_stop(float* avg_ms, int32_t* n_launches) {
  DN_CHECK_ARG(dn::g_prof.cap > 0, "dn_profile_stop: no open profile");
  double tot = 0;
  for (int i = 0; i < dn::g_prof.n; ++i) {
    float ms = 0;
    (void)hipEventSynchronize(dn::g_prof.ev[2 * i + 1]);
    (void)hipEventElapsedTime(&ms, dn::g_prof.ev[2 * i], dn::g_prof.ev[2 * i + 1]);
    tot += ms;
  }
  if (avg_ms) *avg_ms = dn::g_prof.n ? (float)(tot / dn::g_prof.n) : 0.f;
  if (n_launches) *n_launches = dn::g_prof.n;
  for (int i = 0; i < 2 * dn::g_prof.cap; ++i) (void)hipEventDestroy(dn::g_prof.ev[i]);
  delete[] dn::g_prof.ev;
  dn::g_prof = dn::LaunchProfile();
  return DN_OK;
}
