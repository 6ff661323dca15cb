// Host-side orchestration of the DiffNorm hot path on top of the op-level C ABI: sub-model
// descriptors (WaveNet, conditionable transformer), the workspace bump allocator, and the packed
// weight order shared with diffnorm_amd/packing.py.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "../../include/diffnorm_hip.h"

namespace dn {

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline int padk(int c) { return round_up(c, 64); }    // channel width as a K dimension / row stride
inline int padn(int c) { return round_up(c, 128); }   // packed weight rows

inline const void* eoff(const void* p, size_t elems, int es) { return static_cast<const char*>(p) + elems * es; }
inline void* eoff(void* p, size_t elems, int es) { return static_cast<char*>(p) + elems * es; }
inline int esize(int dtype) { return dtype == DN_BF16 || dtype == DN_F16 ? 2 : 4; }

#define DN_TRY(expr)          \
  do {                        \
    int rc__ = (expr);        \
    if (rc__ != DN_OK) return rc__; \
  } while (0)

inline DnGemmParams gemm_base(int dtype, int M, int N, int K, int T) {
  DnGemmParams p;
  memset(&p, 0, sizeof(p));
  p.dtype = dtype;
  p.M = M; p.N = N; p.K = K; p.T = T;
  p.groups = 1;
  p.n_terms = 1;
  p.epilogue = DN_EPI_BIAS;
  p.out_dtype = dtype;
  p.res_dtype = dtype == DN_BF16X3 ? DN_F32 : dtype;  // (an epilogue's residual input is never split rows)
  return p;
}

struct Arena {  // bump allocator over the caller's workspace; base == nullptr only measures
  char* base;
  size_t off, cap;
  void* take(size_t bytes) {
    size_t a = (off + 255) & ~size_t(255);
    off = a + bytes;
    if (!base) return reinterpret_cast<void*>(a + 256);  // non-null dummy while measuring
    return off <= cap ? base + a : nullptr;
  }
};

// WaveNet / WavenetEncoder (latent_module.py:585-617, 1003-1032).  Packed tensors (packing.py):
//   init_W [3][padn(cout)][padk(cin)]   init_b [padk(cout)]
//   conv_W [S][L][3][padn(cout)][padk(cout)]   conv_b [S][L][padk(cout)]
//   res_W  [S][L][padn(cout)][padk(cout)]      res_b  [S][L][padk(cout)]
//   skip_W [L][padn(cout)][padk(cout)]         skip_b [padk(cout)]  (sum over the L blocks)
//   final_W [padn(cout)][padk(cout)]           final_b [padk(cout)]
//   conv_Wkb, res_Wkb: conv_W and res_W once more, K-blocked ([..][padk(cout)/32][padn(cout)][32], bf16; DN_LAYOUT_W_KBLOCKED)
//                      for the 256 x 256 tile, whose fill path moves whole cache lines; any pointer (unused) in f32 mode
struct WavenetW {
  int cin, cout, stacks, layers;
  const void *init_W, *conv_W, *res_W, *skip_W, *final_W, *conv_Wkb, *res_Wkb;
  const float *init_b, *conv_b, *res_b, *skip_b, *final_b;
};
constexpr int kWavenetTensors = 12;

// ConditionableTransformer (latent_module.py:642-706).  Packed tensors:
//   qkv_W [depth][padn(3*hd)][padk(D)]  (rows: to_q ; to_kv)       out_W [depth][padn(D)][padk(hd)]
//   ffin_W [depth][2*padk(inner)][padk(D)] GEGLU-interleaved        ffin_b [depth][2*padk(inner)]
//   ffconv_W [depth][3][padn(inner)][padk(inner)]                   ffconv_b [depth][padk(inner)]
//   ffconv_Wkb: the same weights K-blocked, [depth][3][padk(inner)/32][padn(inner)][32] (bf16; DN_LAYOUT_W_KBLOCKED) -- the
//               form the 256 x 352 tile stages as whole cache lines; any pointer (unused) in f32 mode
//   ffin_Wkb: ffin_W K-blocked, [depth][padk(D)/32][2*padk(inner)][32] (bf16) -- used when the GEGLU projection lands on the
//               256 x 256 tile: the split norm's producer then writes its activations K-blocked too (norm_split = 2)
//   qkv_Wkb: qkv_W K-blocked, [depth][padk(D)/32][padn(3*hd)][32] (bf16) -- layers >= 1 when the projection lands on the 256 x 256
//               tile: the previous layer's residual-closing contraction then writes the attention norm's row * gamma K-blocked
//   ffout_W [depth][padn(D)][padk(inner)]                           ffout_b [depth][padk(D)]
//   g1, g2 [depth][D] learned RMSNorm gammas (NULL when time-conditioned)
//   pred_gamma [D]   pred_W [padn(D)][padk(D)]
struct TransformerW {
  int dim, depth, heads, dim_head, inner;
  const void *qkv_W, *out_W, *ffin_W, *ffconv_W, *ffout_W, *pred_W, *ffconv_Wkb, *ffin_Wkb, *qkv_Wkb;
  const float *ffin_b, *ffconv_b, *ffout_b, *g1, *g2, *pred_gamma;
};
constexpr int kTransformerTensors = 15;

// grouped launch of the row-major weight gradient (wgrad_tn.hip): the blocks of a WaveNet stack in one launch
struct WgTnGroups {
  int groups;
  int64_t dy_gstride, x_gstride, grad_gstride;  // elements between the groups' dY / X / packed gradients
  int shift_by_group;                           // shifts scaled by 2^group
};
int wgrad_tn_launch(const void* dy, int lddy, int cout, const void* const* x, const int* ldx, const int* shift, int n_taps, int cin, int B, int T,
                    int slices, float* part, float* grad, void* stream, int tag, const WgTnGroups* grp);

}  // namespace dn

struct DnEps {
  DnEpsConfig cfg;
  // packed tensors in table order: w_freq, tc_W, tc_b, cond_W, cond_b, init_W, init_b,
  // <wavenet x10>, <transformer x12>, final_W, final_b, pos_table
  const float *w_freq, *tc_W, *tc_b, *cond_b, *init_b, *final_b, *pos_table;
  const void *cond_W, *init_W, *final_W;
  dn::WavenetW wn;
  dn::TransformerW tf;
  int n_cond;  // conditioning columns of a row: 2*padk(dim) per conditioned module
  int n_row;   // row stride of the conditioning table: n_cond, then (split RMSNorm) the depth x (3 hd + 2 padk(inner))
               // columns of beta . W^T for the adaptive norms' consumers
  // conditional variant (kEpsCondTensors; null when cfg.dim_prompt == 0)
  const float *tpc_W, *tpc_b, *null_pc, *lat_pos, *rffin_b, *rffout_b, *rnorm_g, *proj_b;
  const void *null_tok, *proj_W, *rq_W, *rkv_W, *rout_W, *rffin_W, *rffout_W, *cq_W, *ckv_W, *cout_W;
  // hipGraph cache for dn_ddim_loop
  void* graph_exec;
  int graph_B, graph_T;
  void* graph_ws;
  float* graph_x;
  const int32_t* graph_len;
  const float* graph_coef;
  int graph_flags;
  uint64_t graph_seed;  // dn_ddpm_loop: the Philox key baked into the captured step
  void *side_stream, *ev_fork, *ev_join;  // DN_LOOP_SPLIT2: second half-batch stream and its fork/join events
  // DN_LOOP_KEEP_TABLE: the conditioning table built by the previous dn_ddim_loop call on this workspace
  void* table_ws;
  int table_B, table_T, table_split, table_rows;
};
constexpr int kEpsTensors = 7 + dn::kWavenetTensors + dn::kTransformerTensors + 3;
// conditional variant (cfg.dim_prompt > 0), appended to the table: tpc_W (fp32 [padn(C)][padk(P)]), tpc_b [C], null_prompt_cond [C],
// null_prompt_tokens [m][Dp], resampler proj_W [padn(D)][padk(P)], proj_b [Dp], latents + positions fp32 [m][Dp], per resampler layer
// (stacked) q_W [padn(hd)][Dp], kv_W [padn(2hd)][Dp], out_W [padn(D)][hd], ffin_W [2 ip][Dp] (GEGLU-interleaved), ffin_b [2 ip],
// ffout_W [padn(D)][ip], ffout_b [Dp], norm gamma [D]; per transformer layer (stacked) cross-attention q_W, kv_W, out_W.
constexpr int kEpsCondTensors = 18;

struct DnVae {
  DnVaeConfig cfg;
  int n_wave;
  dn::WavenetW enc[4], dec[4];
  dn::TransformerW tf;
  const void* lm_W;
  const float* lm_b;
};
