// Whole-path engine: the eps-predictor `Model`, the speech VAE and the DDIM device loop, expressed as
// sequences of dn_conv_gemm / dn_attention / pointwise launches on one stream.  Nothing here
// allocates or synchronises; every intermediate lives in the caller's workspace (bump-allocated by
// the plan_* functions, which also serve the *_workspace_bytes queries).
#include <algorithm>
#include <new>
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "engine.h"

using namespace dn;

namespace {

// The residual-closing contractions can also emit the next block's RMSNorm (64 x 512 whole-row tile, dn_conv_gemm's
// norm_out).  Measured on [32,512] x dim 512 it loses to 256 x 128 tiles + a standalone norm kernel (one workgroup
// re-reads the whole weight for 64 rows): 6.85 vs 6.63 ms per denoising step.  Off unless DN_FUSE_NORM=1.
inline bool fuse_norm_enabled(int Dp, int dtype) {
  const bool on = option_or(OPT_FUSE_NORM, 0) != 0;  // dn_set_option("fuse_norm", 1)
  return on && Dp <= 512 && dtype != DN_BF16X3;  // (the whole-row tile is not built for split operands)
}

// set by the two-stream sampling loop while it enqueues (or captures) a step: dn_conv_gemm's tile choice counts the twin launch
static thread_local bool g_twin_launches = false;

// Split RMSNorm (DnGemmParams.norm_split / row_ssq): every RMSNorm of the transformer is divided between the residual-
// closing contraction that produces its input and the projection that consumes its output, so the 2*depth+1 norm passes
// over the residual stream disappear (measured: -6 % per denoising step at [32,512] x dim 512).  DN_NO_SPLIT_NORM=1 runs
// the standalone norm kernel instead (A/B timing, and the reference point of the parity tests).
inline bool split_norm_enabled(int Dp, int dtype) {
  return option_or(OPT_NO_SPLIT_NORM, 0) == 0 && Dp % 64 == 0 && !fuse_norm_enabled(Dp, dtype);
}
// DN_BF16X3: tensors that are read by an epilogue or by the attention kernel rather than staged as a contraction operand stay
// plain fp32 (the WaveNet block's residual branch, q / k / v)
inline int side_dtype(int dtype) { return dtype == DN_BF16X3 ? DN_F32 : dtype; }

// DN_KBLOCK: unset = K-blocked buffers where the consuming contraction lands on a tile that gains from them, 0 = never,
// 1 = always (the contraction then runs on a tile that takes them; tests).  Read per call (host side, once per capture).
int kblock_mode() {
  const int v = option(OPT_KBLOCK);  // dn_set_option("kblock", 0 / 1); DN_OPTION_DEFAULT = by tile
  return v == DN_OPT_UNSET ? -1 : (v != 0 ? 1 : 0);
}

// ------------------------------------------------------------------------------------------ WaveNet
struct WaveBufs { void *hw, *resb, *blk0, *blk1, *sk; };

WaveBufs plan_wave(const WavenetW& w, int M, int es, Arena& ar) {
  const size_t one = (size_t)M * padk(w.cout) * es;
  WaveBufs b;
  b.hw = ar.take(one);
  b.resb = ar.take(one * w.layers);
  b.blk0 = ar.take(one * w.layers);
  b.blk1 = ar.take(one * w.layers);
  b.sk = ar.take(one);
  return b;
}

// `fin` carries the destination of the final 1x1 conv (out, ldo, out_dtype, N, epilogue and its extras).
int run_wavenet(const WavenetW& w, int dtype, const void* in, int M, int T, const float* gb, int gb_ld, const WaveBufs& wb,
                DnGemmParams fin, hipStream_t s) {
  const int es = esize(dtype);
  const int cinp = padk(w.cin), cp = padk(w.cout), cn = padn(w.cout), L = w.layers, S = w.stacks;
  const size_t mat = (size_t)cn * cp;
  const int64_t plane = (int64_t)M * cp;
  // The hidden states between the init conv, the res convs and the dilated convs go K-blocked ([cp/32][M][32]) when both of
  // their consumers run on the 256 x 256 tile (large M): its 16-row staging pieces are then whole cache lines (-10 % on the
  // dilated conv, -11 % on the res conv at [32,512]).  The last stack's outputs stay row-major for the skip contraction.
  bool kb = false;
  if (kblock_mode() != 0 && dn::dn_is16(dtype) && w.conv_Wkb && w.res_Wkb) {
    DnGemmParams q = gemm_base(dtype, M, cp, cp, T);
    q.groups = L;
    const int t_res = dn_conv_gemm_tile(&q);
    q.n_terms = 3; q.epilogue = DN_EPI_FILM_GATE;
    kb = kblock_mode() == 1 || (t_res == 3 && dn_conv_gemm_tile(&q) == 3);
  }
  const int a_layout = kb ? (DN_LAYOUT_A_KBLOCKED | DN_LAYOUT_W_KBLOCKED) : 0;
  {  // init conv, k=3, dilation 1 (latent_module.py:596,614 / 1014,1029)
    DnGemmParams p = gemm_base(dtype, M, cp, cinp, T);
    p.n_terms = 3;
    for (int j = 0; j < 3; ++j) {
      p.terms[j].A = in; p.terms[j].lda = cinp; p.terms[j].shift = 2 - j;
      p.terms[j].W = eoff(w.init_W, (size_t)j * cn * cinp, es);
    }
    p.bias = w.init_b; p.out = wb.hw; p.ldo = cp;
    p.out_layout = kb ? DN_LAYOUT_OUT_KBLOCKED : 0;
    DN_TRY(dn_conv_gemm(&p, s));
  }
  for (int st = 0; st < S; ++st) {
    const void* in_s = st == 0 ? wb.hw : ((st - 1) & 1 ? wb.blk1 : wb.blk0);
    const int64_t a_gs = st == 0 ? 0 : plane;  // stack 0 feeds one tensor to all blocks (:570-571)
    void* out_s = (st & 1) ? wb.blk1 : wb.blk0;
    {  // res_conv 1x1 of the L blocks (:510,521)
      DnGemmParams p = gemm_base(dtype, M, cp, cp, T);
      p.groups = L;
      p.terms[0].A = in_s; p.terms[0].lda = cp; p.terms[0].a_gstride = a_gs;
      p.terms[0].W = eoff(kb ? w.res_Wkb : w.res_W, (size_t)st * L * mat, es); p.terms[0].w_gstride = (int64_t)mat;
      p.terms[0].layout = a_layout;
      p.bias = w.res_b + (size_t)st * L * cp; p.bias_gstride = cp;
      p.out = wb.resb; p.ldo = cp; p.out_gstride = plane; p.out_dtype = side_dtype(dtype);
      DN_TRY(dn_conv_gemm(&p, s));
    }
    {  // dilated conv k=3 (dilation 2^block) + FiLM + tanh*sigmoid + residual (:509,523-530)
      DnGemmParams p = gemm_base(dtype, M, cp, cp, T);
      p.groups = L;
      p.n_terms = 3;
      for (int j = 0; j < 3; ++j) {
        p.terms[j].A = in_s; p.terms[j].lda = cp; p.terms[j].a_gstride = a_gs;
        p.terms[j].shift = 2 - j; p.terms[j].shift_by_group = 1;
        p.terms[j].W = eoff(kb ? w.conv_Wkb : w.conv_W, ((size_t)st * L * 3 + j) * mat, es); p.terms[j].w_gstride = (int64_t)(3 * mat);
        p.terms[j].layout = a_layout;
      }
      p.bias = w.conv_b + (size_t)st * L * cp; p.bias_gstride = cp;
      p.epilogue = DN_EPI_FILM_GATE;
      p.res = wb.resb; p.ldr = cp; p.res_gstride = plane; p.res_dtype = side_dtype(dtype);
      if (gb) {
        p.gamma_beta = gb + (size_t)st * L * 2 * cp; p.gb_ld = gb_ld; p.gb_half = cp; p.gb_gstride = 2 * cp;
      }
      p.out = out_s; p.ldo = cp; p.out_gstride = plane;
      p.out_layout = kb && st + 1 < S ? DN_LAYOUT_OUT_KBLOCKED : 0;
      p.pad_ = DN_TAG_WN_DILATED << 8;
      DN_TRY(dn_conv_gemm(&p, s));
    }
  }
  const void* last = ((S - 1) & 1) ? wb.blk1 : wb.blk0;
  {  // sum over blocks of skip_conv(out_i): one contraction with L terms (:511,534,617)
    DnGemmParams p = gemm_base(dtype, M, cp, cp, T);
    p.n_terms = L;
    for (int i = 0; i < L; ++i) {
      p.terms[i].A = eoff(last, (size_t)i * plane, es); p.terms[i].lda = cp;
      p.terms[i].W = eoff(w.skip_W, (size_t)i * mat, es);
    }
    p.bias = w.skip_b; p.out = wb.sk; p.ldo = cp;
    DN_TRY(dn_conv_gemm(&p, s));
  }
  fin.dtype = dtype; fin.M = M; fin.K = cp; fin.T = T; fin.groups = 1; fin.n_terms = 1;
  fin.terms[0].A = wb.sk; fin.terms[0].lda = cp; fin.terms[0].W = w.final_W;
  fin.bias = w.final_b;
  return dn_conv_gemm(&fin, s);
}

// ------------------------------------------------------------------------------------------ transformer
struct TfBufs { void *xn, *qkv, *ao, *gg, *fc; float* ssq; };
// row stride of the sums-of-squares buffer: one float per 64 columns, padded to >= 8 and a multiple of 4 so a consumer
// can fetch a row's partials with two 16-byte loads at kernel start
inline int ssq_ld(int Dp) { const int n = Dp / 64; return n <= 8 ? 8 : (n + 3) / 4 * 4; }

TfBufs plan_tf(const TransformerW& w, int M, int es, Arena& ar) {
  const int hd = w.heads * w.dim_head;
  TfBufs b;
  b.xn = ar.take((size_t)M * padk(w.dim) * es);
  b.qkv = ar.take((size_t)M * 3 * hd * es);
  b.ao = ar.take((size_t)M * hd * es);
  b.gg = ar.take((size_t)M * padk(w.inner) * es);
  b.fc = ar.take((size_t)M * padk(w.inner) * es);
  b.ssq = (float*)ar.take((size_t)M * ssq_ld(padk(w.dim)) * 4);  // split RMSNorm: per-64-column sums of squares
  return b;
}

// Producer half of a split RMSNorm on a RESADD / POSEMB contraction: row * gamma -> xn, per-slab sums of squares -> ssq.
void set_split_norm(DnGemmParams& p, const TfBufs& tb, int Dp, int D, int dtype, const float* gamma, const float* gb, int gb_ld) {
  p.norm_out = tb.xn; p.norm_ld = Dp; p.norm_dtype = dtype; p.norm_D = D;
  p.norm_gamma = gamma; p.norm_gb = gb; p.norm_gb_ld = gb_ld; p.norm_gb_half = Dp;
  p.norm_split = 1; p.norm_ssq = tb.ssq; p.norm_ssq_ld = ssq_ld(Dp);
}
// Consumer half: scale the accumulators by sqrt(D)/|row| and add beta . W^T (rb: the consumer's columns of the row-bias
// block, NULL when the norm has no beta).
void set_row_scale(DnGemmParams& p, const TfBufs& tb, int Dp, int D, const float* rb, int rb_ld) {
  p.row_ssq = tb.ssq; p.row_ssq_ld = ssq_ld(Dp); p.row_ssq_parts = Dp / 64; p.row_D = (float)D;
  p.row_bias = rb; p.row_bias_ld = rb_ld;
}

// Fills the fused-RMSNorm fields of a RESADD / POSEMB contraction (see DnGemmParams.norm_out).
void set_norm(DnGemmParams& p, void* xn, int Dp, int D, int dtype, const float* gamma, const float* gb, int gb_ld) {
  p.norm_out = xn; p.norm_ld = Dp; p.norm_dtype = dtype; p.norm_D = D;
  p.norm_gamma = gamma; p.norm_gb = gb; p.norm_gb_ld = gb_ld; p.norm_gb_half = Dp;
}

// gamma / conditioning row of the norm in front of (layer l, sub-block j of nj): j = 0 attention, nj - 1 feed-forward (nj = 3: 1 is the
// cross-attention block of the prompt-conditioned model); l == depth selects to_pred's norm.
struct NormSrc { const float* gamma; const float* gb; };
NormSrc norm_src(const TransformerW& w, const float* gb, int l, int j, int nj = 2) {
  const int D = w.dim, Dp = padk(D);
  if (l == w.depth) return {w.pred_gamma, nullptr};
  const float* g = j == 0 ? w.g1 : w.g2;
  return {g ? g + (size_t)l * D : nullptr, gb ? gb + (size_t)(nj * l + j) * 2 * Dp : nullptr};
}

// The prompt-conditioned model's extra block per layer (latent_module.py:694-700): cross-attention from the frames to the resampled
// prompt latents, no mask.  kv: [depth][B * n_kv][2 hd] keys | values of every layer (they depend on the prompt only); q: [M, hd].
struct CrossAttn { const void* q_W; const void* out_W; const void* kv; int n_kv; void* q; };

// xres fp32 [M, padk(dim)] is updated in place; `pred` receives to_pred's output.  When the model width fits the
// whole-row tile (padk(dim) <= 512) every RMSNorm is fused into the contraction that produces its input: the caller
// supplies the first one (`xn_ready`: tb.xn already holds layer 0's attention norm) or it runs standalone once.
// `rb` (split RMSNorm with adaptive norms only): fp32 rows [Bc, rb_ld] of beta . W^T, layer l at columns
// l * (3 hd [+ hd] + 2 padk(inner)): first the q/kv projection's, [then the cross-attention query projection's,] then the GEGLU
// projection's (packed column order).  `cx`: the cross-attention block between the two (NULL: the unconditional model).
int run_transformer(const TransformerW& w, int dtype, float* xres, int B, int T, const int32_t* lengths, const float* gb, int gb_ld,
                    const float* rb, int rb_ld, const TfBufs& tb, void* pred, int pred_ld, int pred_dtype, bool xn_ready, hipStream_t s,
                    const CrossAttn* cx = nullptr) {
  const int es = esize(dtype), M = B * T;
  const int D = w.dim, Dp = padk(D), Dn = padn(D), hd = w.heads * w.dim_head, ip = padk(w.inner), in_n = padn(w.inner);
  const bool fuse = fuse_norm_enabled(Dp, dtype);
  const bool split = split_norm_enabled(Dp, dtype);
  const int nj = cx ? 3 : 2, rb_layer = 3 * hd + (cx ? hd : 0) + 2 * ip;
  bool scaled = split && xn_ready;  // tb.xn holds row*gamma + tb.ssq its sums of squares (else: the finished norm)
  auto standalone_norm = [&](int l, int j) -> int {
    const NormSrc ns = norm_src(w, gb, l, j, nj);
    return dn_rmsnorm(xres, Dp, tb.xn, Dp, dtype, M, D, T, ns.gamma, ns.gb, gb_ld, Dp, s);
  };
  if (!((fuse || split) && xn_ready)) DN_TRY(standalone_norm(0, 0));
  // The GEGLU projection (K = dim: 16 K-tiles of 64-byte row pieces on the 256 x 256 tile) reads the feed-forward norm's output and
  // nothing else does: with the split norm its producer (the attention-out contraction) can write row * gamma K-blocked, and the
  // projection then stages whole cache lines from it and from a K-blocked copy of its weights.
  const int mid2 = dn::dn_is16(dtype) && M >= 2048 ? option_or(OPT_MID2, 0) : 0;  // option mid2: the K = dim projections on the 256 x 128 two-workgroup tile
  bool geglu_kb = false;
  static const bool geglu_kb_off = getenv("DN_GEGLU_KB") && atoi(getenv("DN_GEGLU_KB")) == 0;  // A/B timing
  if (!geglu_kb_off && split && !fuse && kblock_mode() != 0 && dn::dn_is16(dtype) && w.ffin_Wkb && Dp % 32 == 0) {
    DnGemmParams q = gemm_base(dtype, M, ip, Dp, T);
    q.epilogue = DN_EPI_GEGLU;
    if (mid2 & 2) q.pad_ |= 9 << 16;
    const int gt = dn_conv_gemm_tile(&q);
    geglu_kb = kblock_mode() == 1 || gt == 3 || gt == 9;
  }
  // The same for the q/kv projection of layers >= 1 (layer 0's norm comes from the caller, row-major): its activations are the
  // attention norm's row * gamma, written by the previous layer's feed-forward-out contraction and read by nothing else.
  bool qkv_kb = false;
  static const bool qkv_kb_off = getenv("DN_QKV_KB") && atoi(getenv("DN_QKV_KB")) == 0;  // A/B timing
  if (!qkv_kb_off && split && !fuse && kblock_mode() != 0 && dn::dn_is16(dtype) && w.qkv_Wkb && Dp % 32 == 0) {
    DnGemmParams q = gemm_base(dtype, M, 3 * hd, Dp, T);
    if (mid2 & 1) q.pad_ |= 9 << 16;
    const int qt = dn_conv_gemm_tile(&q);
    qkv_kb = kblock_mode() == 1 || qt == 3 || qt == 9;
  }
  // option qkv_192: the projection's 3 x 512 columns are 8 x 192 but 6 x 256 -- at [32,512] 512 tiles of 256 x 192 are two full rounds
  // of the chip, 384 tiles of 256 x 256 one and a half
  const bool qkv_192 = option_or(OPT_QKV_192, 0) != 0 && dn::dn_is16(dtype) && (3 * hd) % 192 == 0 && M >= 2048;
  for (int l = 0; l < w.depth; ++l) {
    {  // to_q ; to_kv in one contraction (:930-931,945)
      DnGemmParams p = gemm_base(dtype, M, 3 * hd, Dp, T);
      p.terms[0].A = tb.xn; p.terms[0].lda = Dp; p.terms[0].W = eoff(w.qkv_W, (size_t)l * padn(3 * hd) * Dp, es);
      if (qkv_kb && l >= 1) {
        p.terms[0].W = eoff(w.qkv_Wkb, (size_t)l * padn(3 * hd) * Dp, es);
        p.terms[0].layout = DN_LAYOUT_A_KBLOCKED | DN_LAYOUT_W_KBLOCKED;
      }
      p.out = tb.qkv; p.ldo = 3 * hd; p.out_dtype = side_dtype(dtype);
      if (qkv_192) p.pad_ |= 8 << 16;
      else if (mid2 & 1) p.pad_ |= 9 << 16;
      if (scaled) set_row_scale(p, tb, Dp, D, rb ? rb + (size_t)l * rb_layer : nullptr, rb_ld);
      DN_TRY(dn_conv_gemm(&p, s));
    }
    {
      DnAttnParams a;
      memset(&a, 0, sizeof(a));
      a.q = tb.qkv; a.k = eoff(tb.qkv, hd, es); a.v = eoff(tb.qkv, 2 * hd, es); a.out = tb.ao;
      a.ldq = a.ldk = a.ldv = 3 * hd; a.ldo = hd;
      a.B = B; a.T = T; a.heads = w.heads; a.dim_head = w.dim_head; a.dtype = dtype; a.lengths = lengths;
      a.scale = 1.0f / sqrtf((float)w.dim_head);
      DN_TRY(dn_attention(&a, s));
    }
    {  // to_out + residual (:932,692) [+ the feed-forward block's norm (:703)]
      DnGemmParams p = gemm_base(dtype, M, Dp, hd, T);
      p.terms[0].A = tb.ao; p.terms[0].lda = hd; p.terms[0].W = eoff(w.out_W, (size_t)l * Dn * hd, es);
      p.epilogue = DN_EPI_RESADD; p.res = xres; p.ldr = Dp; p.out = xres; p.ldo = Dp; p.out_dtype = DN_F32;
      const NormSrc ns = norm_src(w, gb, l, 1, nj);
      if (fuse) set_norm(p, tb.xn, Dp, D, dtype, ns.gamma, ns.gb, gb_ld);
      else if (split) set_split_norm(p, tb, Dp, D, dtype, ns.gamma, ns.gb, gb_ld);
      if (geglu_kb && !cx) p.norm_split = 2;  // row * gamma goes out K-blocked: its only reader is the GEGLU projection below
      DN_TRY(dn_conv_gemm(&p, s));
    }
    if (!fuse && !split) DN_TRY(standalone_norm(l, 1));
    scaled = split;
    if (cx) {  // cross-attention to the resampled prompt latents, no mask (:694-700)
      {
        DnGemmParams p = gemm_base(dtype, M, hd, Dp, T);
        p.terms[0].A = tb.xn; p.terms[0].lda = Dp; p.terms[0].W = eoff(cx->q_W, (size_t)l * padn(hd) * Dp, es);
        p.out = cx->q; p.ldo = hd; p.out_dtype = side_dtype(dtype);
        if (scaled) set_row_scale(p, tb, Dp, D, rb ? rb + (size_t)l * rb_layer + 3 * hd : nullptr, rb_ld);
        DN_TRY(dn_conv_gemm(&p, s));
      }
      {
        const void* kv = eoff(cx->kv, (size_t)l * B * cx->n_kv * 2 * hd, esize(side_dtype(dtype)));
        DnAttnParams a;
        memset(&a, 0, sizeof(a));
        a.q = cx->q; a.k = kv; a.v = eoff(kv, hd, esize(side_dtype(dtype))); a.out = tb.ao;
        a.ldq = hd; a.ldk = a.ldv = 2 * hd; a.ldo = hd;
        a.B = B; a.T = T; a.Tk = cx->n_kv; a.heads = w.heads; a.dim_head = w.dim_head; a.dtype = dtype; a.lengths = nullptr;
        a.scale = 1.0f / sqrtf((float)w.dim_head);
        DN_TRY(dn_attention(&a, s));
      }
      {
        DnGemmParams p = gemm_base(dtype, M, Dp, hd, T);
        p.terms[0].A = tb.ao; p.terms[0].lda = hd; p.terms[0].W = eoff(cx->out_W, (size_t)l * Dn * hd, es);
        p.epilogue = DN_EPI_RESADD; p.res = xres; p.ldr = Dp; p.out = xres; p.ldo = Dp; p.out_dtype = DN_F32;
        const NormSrc ns = norm_src(w, gb, l, 2, nj);
        if (fuse) set_norm(p, tb.xn, Dp, D, dtype, ns.gamma, ns.gb, gb_ld);
        else if (split) set_split_norm(p, tb, Dp, D, dtype, ns.gamma, ns.gb, gb_ld);
        if (geglu_kb) p.norm_split = 2;
        DN_TRY(dn_conv_gemm(&p, s));
      }
      if (!fuse && !split) DN_TRY(standalone_norm(l, 2));
    }
    // CausalConv1d(inner, inner, 3) (:894), set up first: when it runs on one of the two 256-row tiles its operands go K-blocked --
    // the GEGLU projection writes its output that way and the weights come from their K-blocked copy (the tile then stages
    // 1 KiB pieces of whole cache lines instead of sixteen half-lines: -5 % on this contraction; DN_KBLOCK=0 disables)
    DnGemmParams pc = gemm_base(dtype, M, ip, ip, T);
    pc.n_terms = 3;
    for (int j = 0; j < 3; ++j) {
      pc.terms[j].A = tb.gg; pc.terms[j].lda = ip; pc.terms[j].shift = 2 - j;
      pc.terms[j].W = eoff(w.ffconv_W, ((size_t)l * 3 + j) * in_n * ip, es);
    }
    pc.bias = w.ffconv_b + (size_t)l * ip; pc.out = tb.fc; pc.ldo = ip;
    pc.pad_ = (DN_TAG_FFN_CONV << 8) | (g_twin_launches ? 128 : 0);  // (bit 7: an identical half-batch launch runs beside this one)
    const int conv_tile = dn_conv_gemm_tile(&pc);  // 4 = 256 x 352 (the eps-predictor's width), 3 = 256 x 256 (the VAE's)
    const bool kblocked = kblock_mode() != 0 && dn::dn_is16(dtype) && w.ffconv_Wkb && (kblock_mode() == 1 || conv_tile == 4 || conv_tile == 3);
    if (kblocked)
      for (int j = 0; j < 3; ++j) {
        pc.terms[j].W = eoff(w.ffconv_Wkb, ((size_t)l * 3 + j) * in_n * ip, es);
        pc.terms[j].layout = DN_LAYOUT_A_KBLOCKED | DN_LAYOUT_W_KBLOCKED;
      }
    {  // Linear(D -> 2*inner) + GEGLU (:899,881-884)
      DnGemmParams p = gemm_base(dtype, M, ip, Dp, T);
      p.terms[0].A = tb.xn; p.terms[0].lda = Dp; p.terms[0].W = eoff(w.ffin_W, (size_t)l * 2 * ip * Dp, es);
      if (geglu_kb) {
        p.terms[0].W = eoff(w.ffin_Wkb, (size_t)l * 2 * ip * Dp, es);
        p.terms[0].layout = DN_LAYOUT_A_KBLOCKED | DN_LAYOUT_W_KBLOCKED;
      }
      p.bias = w.ffin_b + (size_t)l * 2 * ip;
      p.epilogue = DN_EPI_GEGLU; p.out = tb.gg; p.ldo = ip;
      if (mid2 & 2) p.pad_ |= 9 << 16;
      p.out_layout = kblocked ? DN_LAYOUT_OUT_KBLOCKED : 0;
      if (scaled) set_row_scale(p, tb, Dp, D, rb ? rb + (size_t)l * rb_layer + 3 * hd + (cx ? hd : 0) : nullptr, rb_ld);
      DN_TRY(dn_conv_gemm(&p, s));
    }
    DN_TRY(dn_conv_gemm(&pc, s));
    {  // Linear(inner -> D) + residual (:902,704) [+ the next layer's attention norm (:691) or to_pred's norm (:677)]
      DnGemmParams p = gemm_base(dtype, M, Dp, ip, T);
      p.terms[0].A = tb.fc; p.terms[0].lda = ip; p.terms[0].W = eoff(w.ffout_W, (size_t)l * Dn * ip, es);
      p.bias = w.ffout_b + (size_t)l * Dp;
      p.epilogue = DN_EPI_RESADD; p.res = xres; p.ldr = Dp; p.out = xres; p.ldo = Dp; p.out_dtype = DN_F32;
      const NormSrc ns = norm_src(w, gb, l + 1, 0, nj);
      if (fuse) set_norm(p, tb.xn, Dp, D, dtype, ns.gamma, ns.gb, gb_ld);
      else if (split) set_split_norm(p, tb, Dp, D, dtype, ns.gamma, ns.gb, gb_ld);
      if (qkv_kb && l + 1 < w.depth) p.norm_split = 2;  // the next layer's q/kv projection reads it K-blocked (to_pred: row-major)
      DN_TRY(dn_conv_gemm(&p, s));
    }
    if (!fuse && !split) DN_TRY(standalone_norm(l + 1, 0));
  }
  // to_pred = RMSNorm(gamma) + Linear(D, D, no bias) (:676-679); its norm came out of the last contraction above
  DnGemmParams p = gemm_base(dtype, M, Dp, Dp, T);
  p.terms[0].A = tb.xn; p.terms[0].lda = Dp; p.terms[0].W = w.pred_W;
  p.out = pred; p.ldo = pred_ld; p.out_dtype = pred_dtype;
  if (pred_ld < Dp) p.N = pred_ld;  // dense fp32 destination narrower than the padded width
  if (split) set_row_scale(p, tb, Dp, D, nullptr, 0);  // learned gamma, no beta
  return dn_conv_gemm(&p, s);
}

const void* const* take_wavenet(WavenetW& w, const void* const* t) {
  w.init_W = t[0]; w.init_b = (const float*)t[1]; w.conv_W = t[2]; w.conv_b = (const float*)t[3];
  w.res_W = t[4]; w.res_b = (const float*)t[5]; w.skip_W = t[6]; w.skip_b = (const float*)t[7];
  w.final_W = t[8]; w.final_b = (const float*)t[9];
  w.conv_Wkb = t[10]; w.res_Wkb = t[11];
  return t + kWavenetTensors;
}

const void* const* take_transformer(TransformerW& w, const void* const* t) {
  w.qkv_W = t[0]; w.out_W = t[1]; w.ffin_W = t[2]; w.ffin_b = (const float*)t[3]; w.ffconv_W = t[4];
  w.ffconv_b = (const float*)t[5]; w.ffout_W = t[6]; w.ffout_b = (const float*)t[7];
  w.g1 = (const float*)t[8]; w.g2 = (const float*)t[9]; w.pred_gamma = (const float*)t[10]; w.pred_W = t[11];
  w.ffconv_Wkb = t[12]; w.ffin_Wkb = t[13]; w.qkv_Wkb = t[14];
  return t + kTransformerTensors;
}

int check_dims(const char* who, int dtype, int dim, int heads, int dim_head, int layers) {
  DN_CHECK_ARG(dtype == DN_F32 || dtype == DN_BF16 || dtype == DN_BF16X3 || dtype == DN_F16, "%s: bad dtype %d", who, dtype);
  DN_CHECK_ARG(dim > 0 && dim % 4 == 0, "%s: dim=%d must be a positive multiple of 4", who, dim);
  DN_CHECK_ARG((heads * dim_head) % 64 == 0, "%s: heads*dim_head=%d must be a multiple of 64", who, heads * dim_head);
  DN_CHECK_ARG(layers >= 1 && layers <= DN_MAX_TERMS, "%s: wavenet layers=%d must be in 1..%d", who, layers, DN_MAX_TERMS);
  return DN_OK;
}

// ------------------------------------------------------------------------------------------ eps plan
struct EpsBufs {
  float *cond, *gb, *xres;
  void *xin, *h0, *tp, *gbh;  // gbh: the conditioning rows in the arithmetic dtype (A operand of the beta . W^T contraction)
  WaveBufs wv;
  TfBufs tf;
};

EpsBufs plan_eps(const DnEps* m, int B, int T, int Bt, Arena& ar) {
  const int es = esize(m->cfg.dtype), M = B * T;
  const int C = m->cfg.dim * m->cfg.cond_mult, Dp = padk(m->cfg.dim), zp = padk(m->cfg.latent);
  EpsBufs b;
  b.cond = (float*)ar.take((size_t)Bt * C * 4);
  b.gb = (float*)ar.take((size_t)Bt * m->n_row * 4);
  b.gbh = m->cfg.dtype == DN_F32 ? nullptr : ar.take((size_t)Bt * m->n_cond * es);
  b.xin = ar.take((size_t)M * zp * es);
  b.h0 = ar.take((size_t)M * Dp * es);
  b.wv = plan_wave(m->wn, M, es, ar);
  b.xres = (float*)ar.take((size_t)M * Dp * 4);
  b.tf = plan_tf(m->tf, M, es, ar);
  b.tp = ar.take((size_t)M * Dp * es);
  return b;
}

__global__ void fill_t_kernel(int32_t* t, int n, const int32_t* counter) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) t[i] = *counter;
}
__global__ void set_counter_kernel(int32_t* counter, int v) { *counter = v; }
__global__ void dec_counter_kernel(int32_t* counter) { *counter -= 1; }

}  // namespace

// =========================================================================================== eps
extern "C" int dn_eps_create(const DnEpsConfig* cfg, const void* const* weights, int32_t n_weights, DnEps** out) {
  DN_CHECK_ARG(cfg && weights && out, "dn_eps_create: null argument");
  const bool cond_model = cfg->dim_prompt > 0;
  const int expect_tensors = kEpsTensors + (cond_model ? kEpsCondTensors : 0);
  DN_CHECK_ARG(n_weights == expect_tensors, "dn_eps_create: expected %d packed tensors, got %d", expect_tensors, n_weights);
  DN_CHECK_ARG(!cond_model || (cfg->dim_prompt % 4 == 0 && cfg->num_latents >= 1 && cfg->resampler_depth >= 1),
               "dn_eps_create: conditional model needs dim_prompt %% 4 == 0, num_latents >= 1, resampler_depth >= 1");
  DN_TRY(check_dims("dn_eps_create", cfg->dtype, cfg->dim, cfg->heads, cfg->dim_head, cfg->wn_layers));
  DN_CHECK_ARG(cfg->dim % 2 == 0 && (cfg->dim * cfg->cond_mult) % 64 == 0, "dn_eps_create: dim*cond_mult must be a multiple of 64");
  DN_CHECK_ARG(cfg->latent % 4 == 0 && cfg->latent > 0, "dn_eps_create: latent=%d must be a multiple of 4", cfg->latent);
  for (int i = 0; i < n_weights; ++i) DN_CHECK_ARG(weights[i] != nullptr, "dn_eps_create: packed tensor %d is null", i);
  DnEps* m = new (std::nothrow) DnEps();
  DN_CHECK_ARG(m != nullptr, "dn_eps_create: out of host memory");
  memset(m, 0, sizeof(*m));
  m->cfg = *cfg;
  const void* const* t = weights;
  m->w_freq = (const float*)t[0]; m->tc_W = (const float*)t[1]; m->tc_b = (const float*)t[2];
  m->cond_W = t[3]; m->cond_b = (const float*)t[4]; m->init_W = t[5]; m->init_b = (const float*)t[6];
  t += 7;
  m->wn.cin = m->wn.cout = cfg->dim; m->wn.stacks = cfg->wn_stacks; m->wn.layers = cfg->wn_layers;
  t = take_wavenet(m->wn, t);
  m->tf.dim = cfg->dim; m->tf.depth = cfg->depth; m->tf.heads = cfg->heads; m->tf.dim_head = cfg->dim_head;
  m->tf.inner = (int)((double)cfg->dim * 4 * 2 / 3);  // int(dim*mult*2/3), latent_module.py:888
  t = take_transformer(m->tf, t);
  m->tf.g1 = m->tf.g2 = nullptr;  // time-conditioned norms carry no learned gamma (:662-663)
  m->final_W = t[0]; m->final_b = (const float*)t[1]; m->pos_table = (const float*)t[2];
  m->n_cond = (cfg->wn_stacks * cfg->wn_layers + (cond_model ? 3 : 2) * cfg->depth) * 2 * padk(cfg->dim);
  m->n_row = m->n_cond + cfg->depth * (3 * cfg->heads * cfg->dim_head + 2 * padk(m->tf.inner));
  if (cond_model) {
    const void* const* c = t + 3;
    m->tpc_W = (const float*)c[0]; m->tpc_b = (const float*)c[1]; m->null_pc = (const float*)c[2]; m->null_tok = c[3];
    m->proj_W = c[4]; m->proj_b = (const float*)c[5]; m->lat_pos = (const float*)c[6];
    m->rq_W = c[7]; m->rkv_W = c[8]; m->rout_W = c[9]; m->rffin_W = c[10]; m->rffin_b = (const float*)c[11];
    m->rffout_W = c[12]; m->rffout_b = (const float*)c[13]; m->rnorm_g = (const float*)c[14];
    m->cq_W = c[15]; m->ckv_W = c[16]; m->cout_W = c[17];
  }
  *out = m;
  return DN_OK;
}

extern "C" void dn_eps_destroy(DnEps* m) {
  if (!m) return;
  if (m->graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)m->graph_exec);
  if (m->ev_fork) (void)hipEventDestroy((hipEvent_t)m->ev_fork);
  if (m->ev_join) (void)hipEventDestroy((hipEvent_t)m->ev_join);
  if (m->side_stream) (void)hipStreamDestroy((hipStream_t)m->side_stream);
  delete m;
}

static size_t eps_ws_core(const DnEps* m, int B, int T) {
  Arena ar{nullptr, 0, 0};
  (void)plan_eps(m, B, T, B, ar);
  return ar.off + 256;
}

extern "C" size_t dn_eps_workspace_bytes(const DnEps* m, int32_t B, int32_t T) {
  if (!m || B <= 0 || T <= 0) return 0;
  return eps_ws_core(m, B, T);
}

namespace {

// Split RMSNorm: the beta of an adaptive norm reaches its consumer as beta . W^T, which depends on the conditioning row only -- one
// grouped contraction per consumer (one group per layer) writes it to rb [n, rb_ld]: per layer [q/kv columns | cross-attention query
// columns (prompt-conditioned model) | GEGLU columns (packed order)].  gb: the n conditioning rows (fp32, row stride gb_ld); gbh:
// room for them in the arithmetic dtype (unused in fp32).
int eps_beta_rows(const DnEps* m, const float* gb, int gb_ld, int n, void* gbh, float* rb, int rb_ld, hipStream_t s) {
  const int Dp = padk(m->cfg.dim), dtype = m->cfg.dtype, es = esize(dtype);
  const TransformerW& w = m->tf;
  const bool cx = m->cfg.dim_prompt > 0;
  const int hd = w.heads * w.dim_head, ip = padk(w.inner), nj = cx ? 3 : 2, rb_layer = 3 * hd + (cx ? hd : 0) + 2 * ip;
  const size_t tf_off = (size_t)m->cfg.wn_stacks * m->cfg.wn_layers * 2 * Dp;  // first transformer norm's [gamma ; beta]
  const void* A = gb;
  int lda = gb_ld;
  if (dtype != DN_F32) {  // operands in the arithmetic dtype
    DN_TRY(dn_convert_rows(gb, DN_F32, gb_ld, gbh, dtype, m->n_cond, n, m->n_cond, s));
    A = gbh; lda = m->n_cond;
  }
  int col = 0;
  for (int j = 0; j < nj; ++j) {  // j = 0: attention norm -> q/kv projection; nj - 1: feed-forward norm -> GEGLU projection
    const bool last = j == nj - 1;
    const int N = j == 0 ? 3 * hd : (last ? 2 * ip : hd);
    DnGemmParams q = gemm_base(dtype, n, N, Dp, 1);
    q.groups = w.depth;
    q.terms[0].A = eoff(A, tf_off + (size_t)j * 2 * Dp + Dp, es); q.terms[0].lda = lda; q.terms[0].a_gstride = (int64_t)nj * 2 * Dp;
    q.terms[0].W = j == 0 ? w.qkv_W : (last ? w.ffin_W : m->cq_W);
    q.terms[0].w_gstride = j == 0 ? (int64_t)padn(3 * hd) * Dp : (last ? (int64_t)2 * ip * Dp : (int64_t)padn(hd) * Dp);
    q.out = rb + col; q.ldo = rb_ld; q.out_dtype = DN_F32; q.out_gstride = rb_layer;
    DN_TRY(dn_conv_gemm(&q, s));
    col += N;
  }
  return DN_OK;
}

// Conditioning table: rows of [gamma ; beta] for the S*L FiLM blocks and the 2*depth adaptive norms,
// one row per entry of `times`.  Always fp32 (exact-f32 MFMA, fp32 weights): the raw integer timestep
// drives activations of O(100), so this tiny contraction is kept out of the bf16 budget.
int eps_cond_rows(const DnEps* m, const int32_t* times, int n, float* cond, float* gb, void* gbh, hipStream_t s) {
  const int D = m->cfg.dim, C = D * m->cfg.cond_mult, Dp = padk(D), dtype = m->cfg.dtype, es = esize(dtype);
  DN_TRY(dn_time_cond(times, n, m->w_freq, D / 2, m->tc_W, m->tc_b, C, cond, nullptr, DN_F32, C, s));
  DnGemmParams p = gemm_base(DN_F32, n, m->n_cond, C, 1);  // (:507,517) and (:624,637), all at once
  p.terms[0].A = cond; p.terms[0].lda = C; p.terms[0].W = m->cond_W;
  p.bias = m->cond_b; p.out = gb; p.ldo = m->n_row; p.out_dtype = DN_F32;
  DN_TRY(dn_conv_gemm(&p, s));
  if (!split_norm_enabled(Dp, dtype)) return DN_OK;
  return eps_beta_rows(m, gb, m->n_row, n, gbh, gb + m->n_cond, m->n_row, s);
}

// Model.forward after the conditioning (latent_module.py:861-876); gb_ld == 0 -> one row for the batch.
int eps_core(const DnEps* m, const float* x, const float* gb, int gb_ld, const int32_t* lengths, int B, int T, float* eps_out,
             const EpsBufs& b, hipStream_t s) {
  const DnEpsConfig& c = m->cfg;
  const int dtype = c.dtype, M = B * T;
  const int D = c.dim, Dp = padk(D), z = c.latent, zp = padk(z);
  DN_TRY(dn_convert_rows(x, DN_F32, z, b.xin, dtype, zp, M, z, s));
  {  // init_conv 1x1: latent -> dim (:734,864)
    DnGemmParams p = gemm_base(dtype, M, Dp, zp, T);
    p.terms[0].A = b.xin; p.terms[0].lda = zp; p.terms[0].W = m->init_W;
    p.bias = m->init_b; p.out = b.h0; p.ldo = Dp;
    DN_TRY(dn_conv_gemm(&p, s));
  }
  {  // WaveNet; its final 1x1 conv also adds the positional embedding and opens the fp32 residual stream
    DnGemmParams fin = gemm_base(dtype, M, Dp, Dp, T);
    fin.epilogue = DN_EPI_POSEMB; fin.pos_table = m->pos_table; fin.pos_ld = Dp; fin.lengths = lengths;
    fin.out = b.xres; fin.ldo = Dp; fin.out_dtype = DN_F32;
    // layer 0's attention norm rides on the contraction that opens the residual stream
    const float* gb0 = gb + (size_t)c.wn_stacks * c.wn_layers * 2 * Dp;
    if (fuse_norm_enabled(Dp, dtype)) set_norm(fin, b.tf.xn, Dp, D, dtype, nullptr, gb0, gb_ld);
    else if (split_norm_enabled(Dp, dtype)) set_split_norm(fin, b.tf, Dp, D, dtype, nullptr, gb0, gb_ld);
    DN_TRY(run_wavenet(m->wn, dtype, b.h0, M, T, gb, gb_ld, b.wv, fin, s));
  }
  const float* gb_tf = gb + (size_t)c.wn_stacks * c.wn_layers * 2 * Dp;
  const bool xn_ready = fuse_norm_enabled(Dp, dtype) || split_norm_enabled(Dp, dtype);
  DN_TRY(run_transformer(m->tf, dtype, b.xres, B, T, lengths, gb_tf, gb_ld, gb + m->n_cond, gb_ld, b.tf, b.tp, Dp, dtype, xn_ready, s));
  // final_proj: dim -> latent (:807,875), dense fp32 out
  DnGemmParams p = gemm_base(dtype, M, z, Dp, T);
  p.terms[0].A = b.tp; p.terms[0].lda = Dp; p.terms[0].W = m->final_W;
  p.bias = m->final_b; p.out = eps_out; p.ldo = z; p.out_dtype = DN_F32;
  return dn_conv_gemm(&p, s);
}

__global__ void copy_cond_row_kernel(const float* __restrict__ table, int n_cond, const int32_t* __restrict__ counter,
                                     float* __restrict__ dst) {
  const float4* src = reinterpret_cast<const float4*>(table + (size_t)(*counter) * n_cond);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_cond / 4; i += gridDim.x * 256) reinterpret_cast<float4*>(dst)[i] = src[i];
}

__global__ void iota_kernel(int32_t* t, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) t[i] = i;
}

}  // namespace

extern "C" int dn_eps_forward(DnEps* m, const float* x, const int32_t* t, const int32_t* lengths, int32_t B, int32_t T,
                              int32_t shared_t, float* eps_out, void* workspace, size_t workspace_bytes, void* stream) {
  DN_CHECK_ARG(m && x && t && lengths && eps_out && workspace, "dn_eps_forward: null argument");
  DN_CHECK_ARG(m->cfg.dim_prompt == 0, "dn_eps_forward: this model is conditioned on a prompt: use dn_eps_forward_cond");
  DN_CHECK_ARG(B > 0 && T > 0, "dn_eps_forward: B=%d T=%d", B, T);
  DN_CHECK_ARG(T <= m->cfg.max_pos, "dn_eps_forward: T=%d exceeds the positional table (%d)", T, m->cfg.max_pos);
  DN_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "dn_eps_forward: workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int Bt = shared_t ? 1 : B;
  Arena ar{(char*)workspace, 0, workspace_bytes};
  EpsBufs b = plan_eps(m, B, T, Bt, ar);
  if (ar.off > workspace_bytes) {
    dn_set_error("dn_eps_forward: workspace %zu < required %zu", workspace_bytes, ar.off);
    return DN_EWORKSPACE;
  }
  DN_TRY(eps_cond_rows(m, t, Bt, b.cond, b.gb, b.gbh, s));
  return eps_core(m, x, b.gb, shared_t ? 0 : m->n_row, lengths, B, T, eps_out, b, s);
}

static size_t ddim_extra_bytes(const DnEps* m, int B, int T, int start_step) {
  const size_t C = (size_t)m->cfg.dim * m->cfg.cond_mult;
  return (size_t)B * T * m->cfg.latent * 4 + (size_t)start_step * (m->n_row + C) * 4 + (size_t)start_step * m->n_cond * 4 +
         (size_t)(B + start_step) * 4 + 4096;
}

extern "C" size_t dn_ddim_workspace_bytes(const DnEps* m, int32_t B, int32_t T, int32_t start_step) {
  if (!m || B <= 0 || T <= 0 || start_step < 1) return 0;
  const size_t whole = eps_ws_core(m, B, T);
  const size_t halves = B >= 2 ? eps_ws_core(m, B / 2, T) + eps_ws_core(m, B - B / 2, T) : 0;  // DN_LOOP_SPLIT2
  return (whole > halves ? whole : halves) + ddim_extra_bytes(m, B, T, start_step);
}

int dn_ddpm_step_launch(float* x, const float* eps, int M, int C, int T, const float* table, const int32_t* t, int clip, const float* noise,
                        int64_t noise_row, int t_top, uint64_t seed, hipStream_t stream);  // pointwise.hip

namespace {
// the scheduler update applied after every evaluation of the device loop
struct StepOp {
  bool ddpm = false;        // false: DDIM eta = 0 with `coef` [timesteps, 4]; true: ancestral step with `coef` = table [timesteps, DN_GD_COLS]
  int clip = 0;
  uint64_t seed = 0;
  const float* noise = nullptr;  // injected noise rows (ddpm) or NULL
};
}  // namespace

static int sampler_loop(DnEps* m, float* x, const int32_t* lengths, int32_t B, int32_t T, int32_t start_step, int32_t max_evals,
                        const float* coef, int32_t timesteps, int32_t flags, void* workspace, size_t workspace_bytes, void* stream,
                        const StepOp& op) {
  DN_CHECK_ARG(m && m->cfg.dim_prompt == 0, "dn_ddim_loop: the device loop covers the unconditional model (prompted chains step through dn_eps_forward_cond)");
  int use_graph = flags & DN_LOOP_GRAPH;
  const bool split = (flags & DN_LOOP_SPLIT2) && B >= 2;
  DN_CHECK_ARG(m && x && lengths && coef && workspace, "dn_ddim_loop: null argument");
  DN_CHECK_ARG(start_step >= 1 && start_step <= timesteps - (op.ddpm ? 0 : 1), "dn_ddim_loop: start_step=%d must be in [1, %d]", start_step,
               timesteps - (op.ddpm ? 0 : 1));
  hipStream_t s = (hipStream_t)stream;
  DN_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "dn_ddim_loop: workspace must be 256-byte aligned");
  const int z = m->cfg.latent, M = B * T, C = m->cfg.dim * m->cfg.cond_mult;
  // Two half-batches on two streams (a fork/join inside the captured step): the halves are independent chains, so the
  // fill / drain of one half's launches overlaps the other half's main loops (measured -5 % per step at [32,512]).
  const int B0 = split ? B / 2 : B, B1 = B - B0;
  const size_t core0 = eps_ws_core(m, B0, T), core1 = split ? eps_ws_core(m, B1, T) : 0, core = core0 + core1;
  const size_t need = core + ddim_extra_bytes(m, B, T, start_step);
  if (need > workspace_bytes) {
    dn_set_error("dn_ddim_loop: workspace %zu < required %zu (see dn_ddim_workspace_bytes)", workspace_bytes, need);
    return DN_EWORKSPACE;
  }
  Arena core_ar{(char*)workspace, 0, core0};
  const EpsBufs bufs = plan_eps(m, B0, T, 1, core_ar);
  Arena core_ar1{(char*)workspace + core0, 0, core1};
  EpsBufs bufs1 = bufs;
  if (split) bufs1 = plan_eps(m, B1, T, 1, core_ar1);
  if (split && !m->side_stream) {
    if (hipStreamCreateWithFlags((hipStream_t*)&m->side_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags((hipEvent_t*)&m->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags((hipEvent_t*)&m->ev_join, hipEventDisableTiming) != hipSuccess) {
      dn_set_error("dn_ddim_loop: could not create the side stream");
      return DN_ELAUNCH;
    }
  }
  hipStream_t s2 = (hipStream_t)m->side_stream;
  Arena ar{(char*)workspace + core, 0, workspace_bytes - core};
  // fixed-size state first, so a cached graph stays valid when only start_step changes
  float* eps = (float*)ar.take((size_t)M * z * 4);
  int32_t* tvec = (int32_t*)ar.take((size_t)B * 4);
  int32_t* counter = (int32_t*)ar.take(64);
  float* table = (float*)ar.take((size_t)start_step * m->n_row * 4);  // conditioning rows for t = 0..start_step-1
  float* cond_all = (float*)ar.take((size_t)start_step * C * 4);
  void* table_h = m->cfg.dtype == DN_F32 ? nullptr : ar.take((size_t)start_step * m->n_cond * esize(m->cfg.dtype));
  int32_t* tall = (int32_t*)ar.take((size_t)start_step * 4);
  const int last = (start_step == 1 || op.ddpm) ? 0 : 1;  // DDIM: the loop breaks after the t == 1 update (:1444-1445); p_sample_loop runs t = 0 too
  int n_eval = start_step - last;             // t = start_step-1 ... last
  if (max_evals > 0 && max_evals < n_eval) n_eval = max_evals;  // partial chain (benchmarks, chunked sampling)
  // The 56 conditioning vectors depend only on t: build them for the whole chain once (fp32), so the
  // 117 M conditioning weights are not re-streamed at every step.
  const bool keep = (flags & DN_LOOP_KEEP_TABLE) && m->table_ws == workspace && m->table_B == B && m->table_T == T &&
                    m->table_split == (int)split && m->table_rows >= start_step;
  if (!keep) {
    hipLaunchKernelGGL(iota_kernel, dim3((start_step + 255) / 256), dim3(256), 0, s, tall, start_step);
    DN_TRY(eps_cond_rows(m, tall, start_step, cond_all, table, table_h, s));
    m->table_ws = workspace; m->table_B = B; m->table_T = T; m->table_split = (int)split; m->table_rows = start_step;
  }
  const int noise_top = start_step - 1;  // injected noise: row (noise_top - t) belongs to step t
  g_twin_launches = dn::g_gemm_twin = split;  // tile choice of the half-batch launches (host side, also at graph capture)
  struct TwinReset { ~TwinReset() { g_twin_launches = dn::g_gemm_twin = false; } } twin_reset;
  auto one_step = [&]() -> int {
    hipLaunchKernelGGL(fill_t_kernel, dim3((B + 255) / 256), dim3(256), 0, s, tvec, B, counter);
    hipLaunchKernelGGL(copy_cond_row_kernel, dim3(32), dim3(256), 0, s, table, m->n_row, counter, bufs.gb);
    if (split) {  // fork: the second half runs on the side stream behind the shared conditioning row
      const size_t off = (size_t)B0 * T * z;
      if (hipEventRecord((hipEvent_t)m->ev_fork, s) != hipSuccess || hipStreamWaitEvent(s2, (hipEvent_t)m->ev_fork, 0) != hipSuccess) {
        dn_set_error("dn_ddim_loop: fork failed");
        return DN_ELAUNCH;
      }
      DN_TRY(eps_core(m, x + off, bufs.gb, 0, lengths + B0, B1, T, eps + off, bufs1, s2));
      if (op.ddpm)
        DN_TRY(dn_ddpm_step_launch(x + off, eps + off, B1 * T, z, T, coef, tvec + B0, op.clip, op.noise ? op.noise + off : nullptr, (int64_t)M * z,
                                   noise_top, op.seed ^ 0x9E3779B97F4A7C15ull, s2));  // (the second half draws from its own key)
      else
        DN_TRY(dn_ddim_step(x + off, eps + off, x + off, nullptr, DN_F32, z, B1 * T, z, z, T, coef, tvec + B0, s2));
    }
    DN_TRY(eps_core(m, x, bufs.gb, 0, lengths, B0, T, eps, bufs, s));
    if (op.ddpm)
      DN_TRY(dn_ddpm_step_launch(x, eps, B0 * T, z, T, coef, tvec, op.clip, op.noise, (int64_t)M * z, noise_top, op.seed, s));
    else
      DN_TRY(dn_ddim_step(x, eps, x, nullptr, DN_F32, z, B0 * T, z, z, T, coef, tvec, s));
    if (split) {  // join
      if (hipEventRecord((hipEvent_t)m->ev_join, s2) != hipSuccess || hipStreamWaitEvent(s, (hipEvent_t)m->ev_join, 0) != hipSuccess) {
        dn_set_error("dn_ddim_loop: join failed");
        return DN_ELAUNCH;
      }
    }
    hipLaunchKernelGGL(dec_counter_kernel, dim3(1), dim3(1), 0, s, counter);
    DN_CHECK_LAUNCH("dn_ddim_loop step");
    return DN_OK;
  };
  hipLaunchKernelGGL(set_counter_kernel, dim3(1), dim3(1), 0, s, counter, start_step - 1);
  int done = 0;
  if (!s) use_graph = 0;  // the null stream cannot be captured
  if (use_graph && n_eval > 2) {
    const int gflags = (flags & ~DN_LOOP_KEEP_TABLE) | (op.ddpm ? 1 << 16 : 0) | (op.clip ? 1 << 17 : 0);
    // (an injected-noise chain bakes noise_top into the captured step: never served from the cache)
    const bool cached = m->graph_exec && m->graph_B == B && m->graph_T == T && m->graph_ws == workspace && m->graph_x == x &&
                        m->graph_len == lengths && m->graph_coef == coef && m->graph_flags == gflags && !op.noise &&
                        m->graph_seed == op.seed;
    if (!cached) {
      DN_TRY(one_step());  // eager first step: also settles the per-kernel attributes outside capture
      done = 1;
      if (m->graph_exec) {
        (void)hipGraphExecDestroy((hipGraphExec_t)m->graph_exec);
        m->graph_exec = nullptr;
      }
      hipGraph_t graph = nullptr;
      if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) != hipSuccess) {
        dn_set_error("dn_ddim_loop: hipStreamBeginCapture failed");
        return DN_ELAUNCH;
      }
      int rc = one_step();
      hipError_t e = hipStreamEndCapture(s, &graph);
      if (rc != DN_OK) return rc;
      if (e != hipSuccess || !graph) {
        dn_set_error("dn_ddim_loop: hipStreamEndCapture: %s", hipGetErrorString(e));
        return DN_ELAUNCH;
      }
      hipGraphExec_t exec = nullptr;
      e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (e != hipSuccess) {
        dn_set_error("dn_ddim_loop: hipGraphInstantiate: %s", hipGetErrorString(e));
        return DN_ELAUNCH;
      }
      m->graph_exec = exec; m->graph_B = B; m->graph_T = T; m->graph_ws = workspace; m->graph_x = x;
      m->graph_len = lengths; m->graph_coef = coef; m->graph_flags = op.noise ? -1 : gflags; m->graph_seed = op.seed;
    }
    for (; done < n_eval; ++done) {
      hipError_t e = hipGraphLaunch((hipGraphExec_t)m->graph_exec, s);
      if (e != hipSuccess) {
        dn_set_error("dn_ddim_loop: hipGraphLaunch: %s", hipGetErrorString(e));
        return DN_ELAUNCH;
      }
    }
  } else {
    for (; done < n_eval; ++done) DN_TRY(one_step());
  }
  return n_eval;
}

extern "C" int dn_ddim_loop(DnEps* m, float* x, const int32_t* lengths, int32_t B, int32_t T, int32_t start_step, int32_t max_evals,
                            const float* coef, int32_t timesteps, int32_t flags, void* workspace, size_t workspace_bytes,
                            void* stream) {
  return sampler_loop(m, x, lengths, B, T, start_step, max_evals, coef, timesteps, flags, workspace, workspace_bytes, stream, StepOp());
}

extern "C" int dn_ddpm_loop(DnEps* m, float* x, const int32_t* lengths, int32_t B, int32_t T, int32_t start_step, int32_t max_evals,
                            const float* table, int32_t timesteps, int32_t clip_denoised, uint64_t seed, const float* noise, int32_t flags,
                            void* workspace, size_t workspace_bytes, void* stream) {
  StepOp op;
  op.ddpm = true; op.clip = clip_denoised; op.seed = seed; op.noise = noise;
  return sampler_loop(m, x, lengths, B, T, start_step, max_evals, table, timesteps, flags, workspace, workspace_bytes, stream, op);
}


// =========================================================================================== conditional variant (f3)
namespace dn {
// pooled[b, c] = mean over ALL Tp positions of the masked prompt (masked_fill(~mask, 0) then Reduce 'b n d -> b d' mean,
// latent_module.py:844-845, 764); prompt fp32 [B, Tp, P] -> pooled fp32 [B, Pp] (pad columns zero)
__global__ __launch_bounds__(256) void prompt_pool_kernel(const float* __restrict__ prompt, const int32_t* __restrict__ plen, int B, int Tp, int P,
                                                          int Pp, float* __restrict__ pooled) {
  const int b = blockIdx.x;
  const int n = plen[b] < Tp ? plen[b] : Tp;
  for (int c = threadIdx.x; c < Pp; c += 256) {
    float s = 0.f;
    if (c < P)
      for (int t = 0; t < n; ++t) s += prompt[((int64_t)b * Tp + t) * P + c];
    pooled[(int64_t)b * Pp + c] = s / (float)Tp;
  }
}
// cond2[b] = [time_cond[b] | (drop[b] ? null_prompt_cond : prompt_cond[b])]  (:846-852)
__global__ __launch_bounds__(256) void assemble_cond2_kernel(const float* __restrict__ tc, const float* __restrict__ pc, const float* __restrict__ null_pc,
                                                             const int32_t* __restrict__ drop, int B, int C, float* __restrict__ out) {
  const int64_t n = (int64_t)B * 2 * C;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int b = (int)(i / (2 * C)), c = (int)(i - (int64_t)b * 2 * C);
    out[i] = c < C ? tc[(int64_t)b * C + c] : (drop[b] ? null_pc[c - C] : pc[(int64_t)b * C + c - C]);
  }
}
// gb[b] = table[t[b] - t0] + gbp[b]: the conditioning rows of a step from the chain's time table and the prompt half
__global__ __launch_bounds__(256) void cond_rows_from_table_kernel(const float* __restrict__ table, const int32_t* __restrict__ t, int t0, int n_t,
                                                                   const float* __restrict__ gbp, int B, int n, float* __restrict__ gb) {
  const int64_t tot = (int64_t)B * (n / 4);
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int b = (int)(i / (n / 4)), c = (int)(i - (int64_t)b * (n / 4)) * 4;
    int r = t[b] - t0;
    r = r < 0 ? 0 : (r >= n_t ? n_t - 1 : r);
    const float4 a = *reinterpret_cast<const float4*>(table + (int64_t)r * n + c);
    const float4 p = *reinterpret_cast<const float4*>(gbp + (int64_t)b * n + c);
    *reinterpret_cast<float4*>(gb + (int64_t)b * n + c) = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
  }
}
// dst rows [b, row0 + j] (row stride of a sample: rows_total) <- src rows: sample-indexed [b, j] (src_bstride = rows * ld) or shared
// (src_bstride = 0); optional per-sample override by `alt` (shared rows) where flag[b] != 0.  fp32 or bf16 sources -> dst dtype.
__global__ __launch_bounds__(256) void place_rows_kernel(const void* __restrict__ src, int src_dtype, int64_t src_bstride, const void* __restrict__ alt,
                                                         const int32_t* __restrict__ flag, void* __restrict__ dst, int dst_dtype, int B, int rows,
                                                         int ld, int rows_total, int row0) {
  const int64_t n = (int64_t)B * rows * ld;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % ld);
    const int j = (int)((i / ld) % rows);
    const int b = (int)(i / ((int64_t)ld * rows));
    const bool use_alt = alt && flag && flag[b] != 0;
    const void* sp = use_alt ? alt : src;
    const int64_t so = (use_alt ? 0 : (int64_t)b * src_bstride) + (int64_t)j * ld + c;
    const float v = src_dtype == DN_BF16X3 ? load1_split(sp, so)
                    : dn_is16(src_dtype) ? from_h16(src_dtype, reinterpret_cast<const uint16_t*>(sp)[so]) : reinterpret_cast<const float*>(sp)[so];
    const int64_t o = ((int64_t)b * rows_total + row0 + j) * ld + c;
    if (dst_dtype == DN_BF16X3)
      store1_split(dst, o, v);
    else if (dn_is16(dst_dtype))
      reinterpret_cast<uint16_t*>(dst)[o] = to_h16(dst_dtype, v);
    else
      reinterpret_cast<float*>(dst)[o] = v;
  }
}
__global__ void iota_i32_kernel(int32_t first, int32_t* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = first + i;
}
__global__ void add_const_i32_kernel(const int32_t* __restrict__ a, int32_t c, int32_t cap, int32_t* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (a[i] < cap ? a[i] : cap) + c;
}
}  // namespace dn

namespace {

struct CondBufs {
  float *cond, *pooled, *pc, *cond2, *gb, *gbp, *lat, *xres, *rb;
  void *prompt_act, *ctx, *kvsrc, *lat_act, *rq, *rkv, *rao, *rgg, *c_act, *ckv, *xin, *h0, *cq, *tp, *gbh;
  int32_t* klen;
  WaveBufs wv;
  TfBufs tf;
};

CondBufs plan_eps_cond(const DnEps* m, int B, int T, int Tp, Arena& ar) {
  const DnEpsConfig& c = m->cfg;
  const int es = esize(c.dtype);
  const size_t M = (size_t)B * T, Dp = padk(c.dim), zp = padk(c.latent), C = (size_t)c.dim * c.cond_mult, Pp = padk(c.dim_prompt);
  const size_t hd = (size_t)c.heads * c.dim_head, ip = padk(m->tf.inner), ml = c.num_latents, Lk = ml + Tp;
  CondBufs b;
  memset(&b, 0, sizeof(b));
  b.cond = (float*)ar.take(B * C * 4); b.pooled = (float*)ar.take(B * Pp * 4); b.pc = (float*)ar.take(B * C * 4);
  b.cond2 = (float*)ar.take(B * 2 * C * 4); b.gb = (float*)ar.take((size_t)B * m->n_cond * 4);
  b.gbp = (float*)ar.take((size_t)B * m->n_cond * 4);  // the prompt half of the conditioning rows (+ bias): prompt-only, kept across steps
  b.prompt_act = ar.take((size_t)B * Tp * Pp * es); b.ctx = ar.take((size_t)B * Tp * Dp * es);
  b.kvsrc = ar.take(B * Lk * Dp * es); b.lat = (float*)ar.take(B * ml * Dp * 4); b.lat_act = ar.take(B * ml * Dp * es);
  b.rq = ar.take(B * ml * hd * es); b.rkv = ar.take(B * Lk * 2 * hd * es); b.rao = ar.take(B * ml * hd * es);
  b.rgg = ar.take(B * ml * ip * es); b.c_act = ar.take(B * ml * Dp * es); b.ckv = ar.take((size_t)c.depth * B * ml * 2 * hd * es);
  b.klen = (int32_t*)ar.take((size_t)B * 4);
  b.xin = ar.take(M * zp * es); b.h0 = ar.take(M * Dp * es);
  b.wv = plan_wave(m->wn, (int)M, es, ar);
  b.xres = (float*)ar.take(M * Dp * 4);
  b.tf = plan_tf(m->tf, (int)M, es, ar);
  b.cq = ar.take(M * hd * es); b.tp = ar.take(M * Dp * es);
  b.rb = (float*)ar.take((size_t)B * c.depth * (4 * hd + 2 * ip) * 4);  // split RMSNorm: beta . W^T of every adaptive norm's consumer, per sample
  b.gbh = es == 4 && c.dtype == DN_F32 ? nullptr : ar.take((size_t)B * m->n_cond * es);
  return b;
}

int attn_call(int dtype, const void* q, int ldq, const void* k, const void* v, int ldkv, void* out, int ldo, int B, int T, int Tk, int heads,
              int dim_head, const int32_t* lengths, hipStream_t s) {
  DnAttnParams a;
  memset(&a, 0, sizeof(a));
  a.q = q; a.k = k; a.v = v; a.out = out;
  a.ldq = ldq; a.ldk = a.ldv = ldkv; a.ldo = ldo;
  a.B = B; a.T = T; a.Tk = Tk; a.heads = heads; a.dim_head = dim_head; a.dtype = dtype; a.lengths = lengths;
  a.scale = 1.0f / sqrtf((float)dim_head);
  return dn_attention(&a, s);
}

inline int ew(int64_t n) { return (int)std::min<int64_t>((n + 255) / 256, 4096); }

}  // namespace

extern "C" size_t dn_eps_cond_workspace_bytes(const DnEps* m, int32_t B, int32_t T, int32_t Tp) {
  if (!m || m->cfg.dim_prompt <= 0 || B <= 0 || T <= 0 || Tp <= 0) return 0;
  Arena ar{nullptr, 0, 0};
  (void)plan_eps_cond(m, B, T, Tp, ar);
  return ar.off + 256;
}

extern "C" int dn_eps_forward_cond(DnEps* m, const float* x, const int32_t* t, const int32_t* lengths, const float* prompt,
                                   const int32_t* prompt_lengths, const int32_t* drop, int32_t B, int32_t T, int32_t Tp, float* eps_out,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  return dn_eps_forward_cond_ex(m, x, t, lengths, prompt, prompt_lengths, drop, B, T, Tp, eps_out, workspace, workspace_bytes, 0, nullptr, 0, 0, stream);
}

// The time half of the conditioning rows for timesteps t0 .. t0 + n_t - 1 of a chain: table[i] = W_c[:, :C] . time_cond(t0 + i)
// (no bias: it rides on the prompt half).  fp32 [n_t, n_cond].  workspace: n_t * (C + 1) * 4 bytes (+ 256).
extern "C" size_t dn_eps_cond_time_table_workspace_bytes(const DnEps* m, int32_t n_t) {
  if (!m || n_t <= 0) return 0;
  return (size_t)n_t * ((size_t)m->cfg.dim * m->cfg.cond_mult + 1) * 4 + 512;
}
extern "C" int dn_eps_cond_time_table(DnEps* m, int32_t t0, int32_t n_t, float* table, void* workspace, size_t workspace_bytes, void* stream) {
  DN_CHECK_ARG(m && table && workspace && n_t > 0 && t0 >= 0, "dn_eps_cond_time_table: bad argument");
  DN_CHECK_ARG(m->cfg.dim_prompt > 0, "dn_eps_cond_time_table: the model was created without a prompt branch");
  DN_CHECK_ARG(((uintptr_t)workspace & 255) == 0 && workspace_bytes >= dn_eps_cond_time_table_workspace_bytes(m, n_t) - 256, "dn_eps_cond_time_table: workspace");
  hipStream_t s = (hipStream_t)stream;
  const int D = m->cfg.dim, C = D * m->cfg.cond_mult;
  int32_t* times = (int32_t*)workspace;
  float* cond = (float*)((char*)workspace + (((size_t)n_t * 4 + 255) & ~(size_t)255));
  hipLaunchKernelGGL(dn::iota_i32_kernel, dim3((n_t + 255) / 256), dim3(256), 0, s, t0, times, n_t);
  DN_TRY(dn_time_cond(times, n_t, m->w_freq, D / 2, m->tc_W, m->tc_b, C, cond, nullptr, DN_F32, C, s));
  DnGemmParams p = gemm_base(DN_F32, n_t, m->n_cond, C, 1);
  p.terms[0].A = cond; p.terms[0].lda = C; p.terms[0].W = m->cond_W; p.terms[0].ldw = 2 * C;
  p.out = table; p.ldo = m->n_cond; p.out_dtype = DN_F32;
  DN_TRY(dn_conv_gemm(&p, s));
  DN_CHECK_LAUNCH("dn_eps_cond_time_table");
  return DN_OK;
}

extern "C" int dn_eps_forward_cond_ex(DnEps* m, const float* x, const int32_t* t, const int32_t* lengths, const float* prompt,
                                      const int32_t* prompt_lengths, const int32_t* drop, int32_t B, int32_t T, int32_t Tp, float* eps_out,
                                      void* workspace, size_t workspace_bytes, int32_t flags, const float* time_table, int32_t table_t0,
                                      int32_t table_n, void* stream) {
  const bool reuse_prompt = (flags & DN_COND_REUSE_PROMPT) != 0;
  DN_CHECK_ARG(m && x && t && lengths && prompt && prompt_lengths && drop && eps_out && workspace, "dn_eps_forward_cond: null argument");
  DN_CHECK_ARG(!time_table || table_n > 0, "dn_eps_forward_cond: a time table needs its row count");
  DN_CHECK_ARG(m->cfg.dim_prompt > 0, "dn_eps_forward_cond: the model was created without a prompt branch (cfg.dim_prompt == 0)");
  DN_CHECK_ARG(B > 0 && T > 0 && Tp > 0 && T <= m->cfg.max_pos, "dn_eps_forward_cond: B=%d T=%d Tp=%d", B, T, Tp);
  DN_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "dn_eps_forward_cond: workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  Arena ar{(char*)workspace, 0, workspace_bytes};
  const CondBufs b = plan_eps_cond(m, B, T, Tp, ar);
  if (ar.off > workspace_bytes) {
    dn_set_error("dn_eps_forward_cond: workspace %zu < required %zu", workspace_bytes, ar.off);
    return DN_EWORKSPACE;
  }
  const DnEpsConfig& c = m->cfg;
  const int dtype = c.dtype, es = esize(dtype), M = B * T;
  const int D = c.dim, Dp = padk(D), Dn = padn(D), z = c.latent, zp = padk(z), C = D * c.cond_mult, P = c.dim_prompt, Pp = padk(P);
  const int hd = c.heads * c.dim_head, ip = padk(m->tf.inner), in_n = padn(m->tf.inner), ml = c.num_latents, Lk = ml + Tp, R = c.resampler_depth;
  // ---- conditioning rows: [time cond | pooled-prompt cond] -> FiLM / adaptive-norm [gamma ; beta] (:841-852), fp32.  The
  // projection is linear in its two halves, so the rows are formed as (time half) + (prompt half + bias): the prompt half and
  // everything else that depends on the prompt only -- the resampler and every layer's cross-attention keys / values -- is computed
  // once and kept in the workspace (DN_COND_REUSE_PROMPT: a later call on the same workspace and shapes skips it), and the time
  // half of a whole chain can come from dn_eps_cond_time_table (the 1.1 GB projection is then never streamed inside the loop).
  if (!time_table) DN_TRY(dn_time_cond(t, B, m->w_freq, D / 2, m->tc_W, m->tc_b, C, b.cond, nullptr, DN_F32, C, s));
  if (!reuse_prompt) {
  hipLaunchKernelGGL(dn::prompt_pool_kernel, dim3(B), dim3(256), 0, s, prompt, prompt_lengths, B, Tp, P, Pp, b.pooled);
  {  // to_prompt_cond: Linear(P -> C) + SiLU (:760-764)
    DnGemmParams p = gemm_base(DN_F32, B, C, Pp, 1);
    p.terms[0].A = b.pooled; p.terms[0].lda = Pp; p.terms[0].W = m->tpc_W;
    p.bias = m->tpc_b; p.epilogue = DN_EPI_SILU; p.out = b.pc; p.ldo = C; p.out_dtype = DN_F32;
    DN_TRY(dn_conv_gemm(&p, s));
  }
  hipLaunchKernelGGL(dn::assemble_cond2_kernel, dim3(ew((int64_t)B * 2 * C)), dim3(256), 0, s, b.pc, b.pc, m->null_pc, drop, B, C, b.cond2);  // (second half: pc or null_pc by drop)
  {  // prompt half + bias
    DnGemmParams p = gemm_base(DN_F32, B, m->n_cond, C, 1);
    p.terms[0].A = b.cond2 + C; p.terms[0].lda = 2 * C; p.terms[0].W = reinterpret_cast<const float*>(m->cond_W) + C; p.terms[0].ldw = 2 * C;
    p.bias = m->cond_b; p.out = b.gbp; p.ldo = m->n_cond; p.out_dtype = DN_F32;
    DN_TRY(dn_conv_gemm(&p, s));
  }
  // ---- PerceiverResampler (:416-471): prompt -> ml latents per sample
  DN_TRY(dn_convert_rows(prompt, DN_F32, P, b.prompt_act, dtype, Pp, B * Tp, P, s));
  {  // proj_context
    DnGemmParams p = gemm_base(dtype, B * Tp, Dp, Pp, Tp);
    p.terms[0].A = b.prompt_act; p.terms[0].lda = Pp; p.terms[0].W = m->proj_W;
    p.bias = m->proj_b; p.out = b.ctx; p.ldo = Dp;
    DN_TRY(dn_conv_gemm(&p, s));
  }
  // keys/values of every resampler layer = [latents ; projected prompt]: the prompt rows are placed once
  hipLaunchKernelGGL(dn::place_rows_kernel, dim3(ew((int64_t)B * Tp * Dp)), dim3(256), 0, s, b.ctx, dtype, (int64_t)Tp * Dp, nullptr, nullptr, b.kvsrc,
                     dtype, B, Tp, Dp, Lk, ml);
  hipLaunchKernelGGL(dn::place_rows_kernel, dim3(ew((int64_t)B * ml * Dp)), dim3(256), 0, s, m->lat_pos, DN_F32, (int64_t)0, nullptr, nullptr, b.lat,
                     DN_F32, B, ml, Dp, ml, 0);
  hipLaunchKernelGGL(dn::add_const_i32_kernel, dim3((B + 255) / 256), dim3(256), 0, s, prompt_lengths, ml, Tp, b.klen, B);  // mask = [ones(ml) ; prompt_mask]
  for (int l = 0; l < R; ++l) {
    DN_TRY(dn_convert_rows(b.lat, DN_F32, Dp, b.lat_act, dtype, Dp, B * ml, Dp, s));
    hipLaunchKernelGGL(dn::place_rows_kernel, dim3(ew((int64_t)B * ml * Dp)), dim3(256), 0, s, b.lat_act, dtype, (int64_t)ml * Dp, nullptr, nullptr,
                       b.kvsrc, dtype, B, ml, Dp, Lk, 0);
    {
      DnGemmParams p = gemm_base(dtype, B * ml, hd, Dp, ml);
      p.terms[0].A = b.lat_act; p.terms[0].lda = Dp; p.terms[0].W = eoff(m->rq_W, (size_t)l * padn(hd) * Dp, es);
      p.out = b.rq; p.ldo = hd; p.out_dtype = side_dtype(dtype);
      DN_TRY(dn_conv_gemm(&p, s));
    }
    {
      DnGemmParams p = gemm_base(dtype, B * Lk, 2 * hd, Dp, Lk);
      p.terms[0].A = b.kvsrc; p.terms[0].lda = Dp; p.terms[0].W = eoff(m->rkv_W, (size_t)l * padn(2 * hd) * Dp, es);
      p.out = b.rkv; p.ldo = 2 * hd; p.out_dtype = side_dtype(dtype);
      DN_TRY(dn_conv_gemm(&p, s));
    }
    DN_TRY(attn_call(dtype, b.rq, hd, b.rkv, eoff(b.rkv, hd, es), 2 * hd, b.rao, hd, B, ml, Lk, c.heads, c.dim_head, b.klen, s));
    {
      DnGemmParams p = gemm_base(dtype, B * ml, Dp, hd, ml);
      p.terms[0].A = b.rao; p.terms[0].lda = hd; p.terms[0].W = eoff(m->rout_W, (size_t)l * Dn * hd, es);
      p.epilogue = DN_EPI_RESADD; p.res = b.lat; p.ldr = Dp; p.out = b.lat; p.ldo = Dp; p.out_dtype = DN_F32;
      DN_TRY(dn_conv_gemm(&p, s));
    }
    DN_TRY(dn_convert_rows(b.lat, DN_F32, Dp, b.lat_act, dtype, Dp, B * ml, Dp, s));
    {  // FeedForward without the causal conv: Linear -> GEGLU -> Linear (:887-903)
      DnGemmParams p = gemm_base(dtype, B * ml, ip, Dp, ml);
      p.terms[0].A = b.lat_act; p.terms[0].lda = Dp; p.terms[0].W = eoff(m->rffin_W, (size_t)l * 2 * ip * Dp, es);
      p.bias = m->rffin_b + (size_t)l * 2 * ip; p.epilogue = DN_EPI_GEGLU; p.out = b.rgg; p.ldo = ip;
      DN_TRY(dn_conv_gemm(&p, s));
    }
    {
      DnGemmParams p = gemm_base(dtype, B * ml, Dp, ip, ml);
      p.terms[0].A = b.rgg; p.terms[0].lda = ip; p.terms[0].W = eoff(m->rffout_W, (size_t)l * Dn * ip, es);
      p.bias = m->rffout_b + (size_t)l * Dp;
      p.epilogue = DN_EPI_RESADD; p.res = b.lat; p.ldr = Dp; p.out = b.lat; p.ldo = Dp; p.out_dtype = DN_F32;
      DN_TRY(dn_conv_gemm(&p, s));
    }
  }
  DN_TRY(dn_rmsnorm(b.lat, Dp, b.lat_act, Dp, dtype, B * ml, D, ml, m->rnorm_g, nullptr, 0, 0, s));
  // c = where(drop, null_prompt_tokens, resampled) (:855-859)
  hipLaunchKernelGGL(dn::place_rows_kernel, dim3(ew((int64_t)B * ml * Dp)), dim3(256), 0, s, b.lat_act, dtype, (int64_t)ml * Dp, m->null_tok, drop, b.c_act,
                     dtype, B, ml, Dp, ml, 0);
  {  // keys / values of all cross-attention layers at once (they depend on the prompt only)
    DnGemmParams p = gemm_base(dtype, B * ml, 2 * hd, Dp, ml);
    p.groups = c.depth;
    p.terms[0].A = b.c_act; p.terms[0].lda = Dp; p.terms[0].a_gstride = 0;
    p.terms[0].W = m->ckv_W; p.terms[0].w_gstride = (int64_t)padn(2 * hd) * Dp;
    p.out = b.ckv; p.ldo = 2 * hd; p.out_gstride = (int64_t)B * ml * 2 * hd; p.out_dtype = side_dtype(dtype);
    DN_TRY(dn_conv_gemm(&p, s));
  }
  }  // !reuse_prompt
  // ---- the step's conditioning rows: time half (from the chain's table, or contracted here) + prompt half
  if (time_table) {
    hipLaunchKernelGGL(dn::cond_rows_from_table_kernel, dim3(ew((int64_t)B * m->n_cond / 4)), dim3(256), 0, s, time_table, t, table_t0, table_n, b.gbp, B,
                       m->n_cond, b.gb);
  } else {
    DnGemmParams p = gemm_base(DN_F32, B, m->n_cond, C, 1);
    p.terms[0].A = b.cond; p.terms[0].lda = C; p.terms[0].W = m->cond_W; p.terms[0].ldw = 2 * C;
    p.epilogue = DN_EPI_RESADD; p.res = b.gbp; p.ldr = m->n_cond; p.out = b.gb; p.ldo = m->n_cond; p.out_dtype = DN_F32;
    DN_TRY(dn_conv_gemm(&p, s));
  }
  const bool split = split_norm_enabled(Dp, dtype), xn_ready = split || fuse_norm_enabled(Dp, dtype);
  const int rb_ld = c.depth * (4 * hd + 2 * ip);
  if (split) DN_TRY(eps_beta_rows(m, b.gb, m->n_cond, B, b.gbh, b.rb, rb_ld, s));
  // ---- the eps-predictor proper (:861-876)
  DN_TRY(dn_convert_rows(x, DN_F32, z, b.xin, dtype, zp, M, z, s));
  {
    DnGemmParams p = gemm_base(dtype, M, Dp, zp, T);
    p.terms[0].A = b.xin; p.terms[0].lda = zp; p.terms[0].W = m->init_W;
    p.bias = m->init_b; p.out = b.h0; p.ldo = Dp;
    DN_TRY(dn_conv_gemm(&p, s));
  }
  {
    DnGemmParams fin = gemm_base(dtype, M, Dp, Dp, T);
    fin.epilogue = DN_EPI_POSEMB; fin.pos_table = m->pos_table; fin.pos_ld = Dp; fin.lengths = lengths;
    fin.out = b.xres; fin.ldo = Dp; fin.out_dtype = DN_F32;
    const float* gb0 = b.gb + (size_t)c.wn_stacks * c.wn_layers * 2 * Dp;  // layer 0's attention norm rides on it
    if (fuse_norm_enabled(Dp, dtype)) set_norm(fin, b.tf.xn, Dp, D, dtype, nullptr, gb0, m->n_cond);
    else if (split) set_split_norm(fin, b.tf, Dp, D, dtype, nullptr, gb0, m->n_cond);
    DN_TRY(run_wavenet(m->wn, dtype, b.h0, M, T, b.gb, m->n_cond, b.wv, fin, s));
  }
  // the transformer of the unconditional model with the cross-attention block between attention and feed-forward: every RMSNorm
  // split between the contraction that produces its input and the one that consumes it, K-blocked operands where a tile gains
  const float* gb_tf = b.gb + (size_t)c.wn_stacks * c.wn_layers * 2 * Dp;
  const CrossAttn cx = {m->cq_W, m->cout_W, b.ckv, ml, b.cq};
  DN_TRY(run_transformer(m->tf, dtype, b.xres, B, T, lengths, gb_tf, m->n_cond, split ? b.rb : nullptr, rb_ld, b.tf, b.tp, Dp, dtype, xn_ready, s, &cx));
  DnGemmParams p = gemm_base(dtype, M, z, Dp, T);
  p.terms[0].A = b.tp; p.terms[0].lda = Dp; p.terms[0].W = m->final_W;
  p.bias = m->final_b; p.out = eps_out; p.ldo = z; p.out_dtype = DN_F32;
  DN_TRY(dn_conv_gemm(&p, s));
  DN_CHECK_LAUNCH("dn_eps_forward_cond");
  return DN_OK;
}

// =========================================================================================== VAE
extern "C" int dn_vae_create(const DnVaeConfig* cfg, const void* const* weights, int32_t n_weights, DnVae** out) {
  DN_CHECK_ARG(cfg && weights && out, "dn_vae_create: null argument");
  DN_CHECK_ARG(cfg->n_mults >= 1 && cfg->n_mults <= 4, "dn_vae_create: n_mults=%d", cfg->n_mults);
  const int expect = 2 * cfg->n_mults * kWavenetTensors + kTransformerTensors + 2;
  DN_CHECK_ARG(n_weights == expect, "dn_vae_create: expected %d packed tensors, got %d", expect, n_weights);
  DN_TRY(check_dims("dn_vae_create", cfg->dtype, cfg->dim, cfg->heads, cfg->dim_head, cfg->layers));
  DN_CHECK_ARG(cfg->vocab % 4 == 0 && cfg->z % 4 == 0, "dn_vae_create: vocab and z must be multiples of 4");
  for (int i = 0; i < n_weights; ++i) DN_CHECK_ARG(weights[i] != nullptr, "dn_vae_create: packed tensor %d is null", i);
  DnVae* m = new (std::nothrow) DnVae();
  DN_CHECK_ARG(m != nullptr, "dn_vae_create: out of host memory");
  memset(m, 0, sizeof(*m));
  m->cfg = *cfg;
  m->n_wave = cfg->n_mults;
  const void* const* t = weights;
  int cur = cfg->dim;
  for (int n = 0; n < m->n_wave; ++n) {  // latent_module.py:1053-1065
    WavenetW& w = m->enc[n];
    w.cin = cur; w.cout = cur / cfg->mults[n]; w.stacks = cfg->stacks; w.layers = cfg->layers;
    cur = w.cout;
    t = take_wavenet(w, t);
  }
  DN_CHECK_ARG(cur == 2 * cfg->z, "dn_vae_create: encoder width %d != 2*z (%d)", cur, 2 * cfg->z);
  for (int n = 0; n < m->n_wave; ++n) {  // :1067-1081 (first decoder input halved by the posterior sample)
    WavenetW& w = m->dec[n];
    const int mult = cfg->mults[m->n_wave - 1 - n];
    w.cout = cur * mult; w.cin = n == 0 ? cur / 2 : cur; w.stacks = cfg->stacks; w.layers = cfg->layers;
    cur = w.cout;
    t = take_wavenet(w, t);
  }
  DN_CHECK_ARG(cur == cfg->dim, "dn_vae_create: decoder width %d != dim %d", cur, cfg->dim);
  m->tf.dim = cfg->dim; m->tf.depth = cfg->depth; m->tf.heads = cfg->heads; m->tf.dim_head = cfg->dim_head;
  m->tf.inner = (int)((double)cfg->dim * 4 * 2 / 3);
  t = take_transformer(m->tf, t);
  m->lm_W = t[0]; m->lm_b = (const float*)t[1];
  *out = m;
  return DN_OK;
}

extern "C" void dn_vae_destroy(DnVae* m) { delete m; }

namespace {
struct VaeBufs {
  void *in_act, *mid[2];
  WaveBufs wv[4];
  float *xres, *recon, *logits;
  void* pred_act;
  TfBufs tf;
};

VaeBufs plan_vae_enc(const DnVae* m, int M, Arena& ar) {
  const int es = esize(m->cfg.dtype);
  VaeBufs b;
  memset(&b, 0, sizeof(b));
  b.in_act = ar.take((size_t)M * padk(m->cfg.dim) * es);
  int widest = 0;
  for (int n = 0; n < m->n_wave; ++n) widest = widest > m->enc[n].cout ? widest : m->enc[n].cout;
  b.mid[0] = ar.take((size_t)M * padk(widest) * es);
  b.mid[1] = ar.take((size_t)M * padk(widest) * es);
  for (int n = 0; n < m->n_wave; ++n) b.wv[n] = plan_wave(m->enc[n], M, es, ar);
  return b;
}

VaeBufs plan_vae_dec(const DnVae* m, int M, bool need_recon, bool need_logits, Arena& ar) {
  const int es = esize(m->cfg.dtype), Dp = padk(m->cfg.dim);
  VaeBufs b;
  memset(&b, 0, sizeof(b));
  b.in_act = ar.take((size_t)M * padk(m->cfg.z) * es);
  b.mid[0] = ar.take((size_t)M * Dp * es);
  b.mid[1] = ar.take((size_t)M * Dp * es);
  for (int n = 0; n < m->n_wave; ++n) b.wv[n] = plan_wave(m->dec[n], M, es, ar);
  b.xres = (float*)ar.take((size_t)M * Dp * 4);
  b.tf = plan_tf(m->tf, M, es, ar);
  b.recon = need_recon ? (float*)ar.take((size_t)M * Dp * 4) : nullptr;
  b.pred_act = ar.take((size_t)M * Dp * es);
  b.logits = need_logits ? (float*)ar.take((size_t)M * m->cfg.vocab * 4) : nullptr;
  return b;
}
}  // namespace

extern "C" size_t dn_vae_workspace_bytes(const DnVae* m, int32_t B, int32_t T) {
  if (!m || B <= 0 || T <= 0) return 0;
  Arena a{nullptr, 0, 0}, d{nullptr, 0, 0};
  (void)plan_vae_enc(m, B * T, a);
  (void)plan_vae_dec(m, B * T, true, true, d);
  return (a.off > d.off ? a.off : d.off) + 256;
}

extern "C" int dn_vae_encode_params(DnVae* m, const float* feat, int32_t B, int32_t T, float* params, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  DN_CHECK_ARG(m && feat && params && workspace && B > 0 && T > 0, "dn_vae_encode_params: bad argument");
  DN_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "dn_vae_encode_params: workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int dtype = m->cfg.dtype, M = B * T;
  Arena ar{(char*)workspace, 0, workspace_bytes};
  VaeBufs b = plan_vae_enc(m, M, ar);
  if (ar.off > workspace_bytes) {
    dn_set_error("dn_vae_encode_params: workspace %zu < required %zu", workspace_bytes, ar.off);
    return DN_EWORKSPACE;
  }
  DN_TRY(dn_convert_rows(feat, DN_F32, m->cfg.dim, b.in_act, dtype, padk(m->cfg.dim), M, m->cfg.dim, s));
  const void* cur = b.in_act;
  for (int n = 0; n < m->n_wave; ++n) {
    const WavenetW& w = m->enc[n];
    const bool lastw = n == m->n_wave - 1;
    DnGemmParams fin = gemm_base(dtype, M, lastw ? w.cout : padk(w.cout), padk(w.cout), T);
    if (lastw) {  // posterior parameters [mean ; logvar], dense fp32
      fin.out = params; fin.ldo = w.cout; fin.out_dtype = DN_F32;
    } else {
      fin.out = b.mid[n & 1]; fin.ldo = padk(w.cout); fin.out_dtype = dtype;
    }
    DN_TRY(run_wavenet(w, dtype, cur, M, T, nullptr, 0, b.wv[n], fin, s));
    cur = b.mid[n & 1];
  }
  return DN_OK;
}

extern "C" int dn_vae_decode(DnVae* m, const float* latent, const int32_t* lengths, int32_t B, int32_t T, float* recon, float* logits,
                             int32_t* units, void* workspace, size_t workspace_bytes, void* stream) {
  DN_CHECK_ARG(m && latent && lengths && workspace && B > 0 && T > 0, "dn_vae_decode: bad argument");
  DN_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "dn_vae_decode: workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int dtype = m->cfg.dtype, M = B * T, D = m->cfg.dim, Dp = padk(D), z = m->cfg.z, V = m->cfg.vocab;
  const bool want_lm = logits != nullptr || units != nullptr;
  const bool dense_recon = recon != nullptr && Dp == D;  // to_pred can write the caller's buffer directly
  Arena ar{(char*)workspace, 0, workspace_bytes};
  VaeBufs b = plan_vae_dec(m, M, !dense_recon, want_lm && !logits, ar);
  if (ar.off > workspace_bytes) {
    dn_set_error("dn_vae_decode: workspace %zu < required %zu", workspace_bytes, ar.off);
    return DN_EWORKSPACE;
  }
  DN_TRY(dn_convert_rows(latent, DN_F32, z, b.in_act, dtype, padk(z), M, z, s));
  const void* cur = b.in_act;
  for (int n = 0; n < m->n_wave; ++n) {
    const WavenetW& w = m->dec[n];
    const bool lastw = n == m->n_wave - 1;
    DnGemmParams fin = gemm_base(dtype, M, padk(w.cout), padk(w.cout), T);
    if (lastw) {  // opens the transformer's fp32 residual stream (no positional embedding, :1109-1114)
      fin.out = b.xres; fin.ldo = Dp; fin.out_dtype = DN_F32;
    } else {
      fin.out = b.mid[n & 1]; fin.ldo = padk(w.cout); fin.out_dtype = dtype;
    }
    DN_TRY(run_wavenet(w, dtype, cur, M, T, nullptr, 0, b.wv[n], fin, s));
    cur = b.mid[n & 1];
  }
  float* rec = dense_recon ? recon : b.recon;
  const int rec_ld = dense_recon ? D : Dp;
  DN_TRY(run_transformer(m->tf, dtype, b.xres, B, T, lengths, nullptr, 0, nullptr, 0, b.tf, rec, rec_ld, DN_F32, false, s));
  if (recon && !dense_recon) DN_TRY(dn_convert_rows(rec, DN_F32, Dp, recon, DN_F32, D, M, D, s));
  if (!want_lm) return DN_OK;
  DN_TRY(dn_convert_rows(rec, DN_F32, rec_ld, b.pred_act, dtype, Dp, M, D, s));
  float* lg = logits ? logits : b.logits;
  DnGemmParams p = gemm_base(dtype, M, V, Dp, T);  // decoder_lm (:1096,1115)
  p.terms[0].A = b.pred_act; p.terms[0].lda = Dp; p.terms[0].W = m->lm_W;
  p.bias = m->lm_b; p.out = lg; p.ldo = V; p.out_dtype = DN_F32;
  DN_TRY(dn_conv_gemm(&p, s));
  if (units) DN_TRY(dn_argmax_units(lg, V, M, V, 4, units, s));
  return DN_OK;
}
