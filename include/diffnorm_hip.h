/*
 * diffnorm_hip.h -- C ABI of libdiffnorm_hip.so: the MI355X (gfx950) implementation of DiffNorm's
 * latent-diffusion denoising hot path.
 *
 * The reference (steventan0110/DiffNorm) has no FFI on this path: its boundary is the fairseq Python
 * plugin API and everything below it is stock ATen.  The entry points here are what a binding for the
 * path would attach to; each cites the reference function it replaces (paths relative to
 * fairseq/models/text_to_speech/ in the reference).  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - plain C types only: device pointers, sizes, enums; no torch types.
 *   - every function returns 0 on success or a negative DN_E* code; dn_last_error() gives the text.
 *   - functions taking a `stream` only enqueue work on it (hipStream_t passed as void*); they never
 *     allocate device memory and never synchronise, so they are capturable into a hipGraph.
 *   - activations are channels-last row-major: row m = b*T + t, `ld` elements between rows.
 *   - `dtype` selects the arithmetic of the contractions: DN_BF16 = bf16 MFMA operands with fp32
 *     accumulation (activations that feed a contraction are stored bf16, the transformer residual
 *     stream stays fp32); DN_F32 = exact fp32 MFMA (v_mfma_f32_16x16x4_f32) end to end;
 *     DN_BF16X3 = split-operand bf16: every contraction operand is stored as a bf16 pair (hi, lo) with x ~ hi + lo
 *     (hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits) and a product runs as three bf16 MFMAs into one fp32
 *     accumulator, a_hi b_hi + a_lo b_hi + a_hi b_lo (the dropped a_lo b_lo term is 2^-18 relative), i.e. fp32-class
 *     results (1e-3 budget of north_star's fp32 column) at a third of the bf16 MFMA rate instead of a sixteenth.
 *     Everything that is not a contraction operand is fp32 exactly as in DN_F32 mode.
 *     DN_F16 = IEEE-half MFMA operands (v_mfma_f32_16x16x32_f16: the bf16 issue rate on gfx950) with fp32 accumulation --
 *     DN_BF16 with three more significand bits: the same 2-byte layouts, tiles and schedules, every tensor DN_BF16 stores as
 *     bf16 is stored as half, everything DN_BF16 keeps fp32 stays fp32.  The format ends at 65504: a value beyond it saturates
 *     (the kernels run with the MODE register's FP16_OVFL bit set) instead of becoming inf.  Inference engines only.
 *   - DN_BF16X3 storage ("split rows"): an element takes 4 bytes like fp32 and ld / K / column offsets count elements, but
 *     every group of 32 consecutive elements of a row is laid out as two 64-byte halves of 32 bf16 each: ACTIVATIONS
 *     (everything a kernel of this library writes, and every A operand) store [hi | lo], packed WEIGHTS (W operands)
 *     store [lo | hi].  A K-tile of 128 bytes per row then reads as two bf16 k-steps, and a kernel that walks a row in
 *     64-byte K-tiles meets (w_lo, a_hi) then (w_hi, a_lo): it keeps only a_hi across the pair for its three products.
 *     Row strides, K and column offsets of such tensors are multiples of 32 elements; bases are 128-byte aligned.
 */
#ifndef DIFFNORM_HIP_H
#define DIFFNORM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { DN_F32 = 0, DN_BF16 = 1, DN_BF16X3 = 2, DN_F16 = 3 };

enum {
  DN_OK = 0,
  DN_EINVAL = -1,   /* bad argument (shape/alignment/enum) */
  DN_ELAUNCH = -2,  /* HIP launch or runtime error */
  DN_EWORKSPACE = -3 /* workspace too small */
};

/* epilogues of dn_conv_gemm */
enum {
  DN_EPI_BIAS = 0,      /* out = acc + bias                                               (nn.Linear / CausalConv1d) */
  DN_EPI_SILU = 1,      /* out = silu(acc + bias)                                          latent_module.py:741-745 */
  DN_EPI_GEGLU = 2,     /* out[:, j] = gelu_erf(gate_j) * value_j (latent_module.py:881-884); W rows packed per 16-row tile:
                           packed row 16 t + i = value row 8 t + i (i < 8) or gate row 8 t + i - 8 (i >= 8)        */
  DN_EPI_FILM_GATE = 3, /* h=(acc+bias)[*gamma+beta]; out = tanh(h)*sigmoid(h) + res      latent_module.py:525-530 */
  DN_EPI_RESADD = 4,    /* out = res + acc + bias  (fp32 residual stream)                  latent_module.py:692,704 */
  DN_EPI_POSEMB = 5,    /* out = acc + bias + pe[pos(b,t)]                                 latent_module.py:867-868 */
  DN_EPI_RELU = 6       /* out = relu(acc + bias)   (the S2UT decoder's FFN, fairseq/modules/transformer_layer.py:505-508) */
};

#define DN_MAX_TERMS 8

/* One additive term of a causal-conv contraction: rows of A shifted back by `shift` frames inside
 * each sequence (rows with t < shift read zeros), times the packed weight W[Np][K].              */
typedef struct {
  const void* A;      /* [M, lda] activations, element type = dtype                                */
  const void* W;      /* [Np, K] packed weights (K contiguous), Np = N rounded up to 128           */
  int32_t lda;        /* elements between rows of A                                                 */
  int32_t shift;      /* causal shift in frames (tap j of a k-tap conv: (k-1-j)*dilation); negative = frames AFTER t
                         (zeros past the sequence end): the transposed conv of the backward data path (f2)   */
  int64_t a_gstride;  /* elements added to A per group (0 = all groups share A)                     */
  int64_t w_gstride;  /* elements added to W per group                                              */
  int32_t shift_by_group; /* 1: effective shift = shift * 2^group (WaveNet dilation 2^i)            */
  int32_t layout;     /* DN_LAYOUT_* bits; 0 = row-major A and W as described above                 */
  int32_t ldw;        /* elements between rows of W when it is not K (0 = K): a K-slice of a wider packed matrix -- with `groups`,
                         a_gstride and w_gstride = the slice width this is a split-K contraction whose groups write partial sums */
  int32_t pad_ldw_;
} DnGemmTerm;

/* K-blocked operand layout, bf16 only: [K/32][rows][32] instead of [rows][K] -- the 32 K-elements (64 bytes) a K-tile
 * takes from each row lie next to the same K-tile's elements of the neighbouring rows, so the 16-row pieces the
 * contraction stages are 1 KiB of whole cache lines (row-major pieces are sixteen half-lines; measured 2-4x the
 * L2 -> LDS fill rate, tools/dma_issue_cost.hip) and a causal shift is a plain row offset.  rows = M for A and for
 * out, Np for W.  K-blocked A / W are taken by the 256 x 352 tile and, where the shape does not route there, by the
 * 256 x 256 tile (the contraction then runs on that tile whatever the shape heuristic would have chosen; forcing
 * another tile is DN_EINVAL); dn_conv_gemm_kblocked_ok tells whether the shape routes to the 256 x 352 tile, where the
 * layout pays.  The GEGLU epilogue can emit K-blocked output (out_layout) on every tile.                          */
enum { DN_LAYOUT_A_KBLOCKED = 1, DN_LAYOUT_W_KBLOCKED = 2, DN_LAYOUT_OUT_KBLOCKED = 1 };

/* out[g] = epilogue( sum_terms shift(A_term[g]) @ W_term[g]^T ), g = 0..groups-1.
 * Replaces nn.Linear / CausalConv1d(k=1,3) and the ops the reference runs after them
 * (latent_module.py:476-488, 513-536, 613-617, 887-903, 930-950).                               */
typedef struct {
  DnGemmTerm terms[DN_MAX_TERMS];
  int32_t n_terms;
  int32_t dtype;       /* DN_F32 | DN_BF16 | DN_F16 | DN_BF16X3: element type of A and W            */
  int32_t M, N, K;     /* rows (B*T), stored output columns (multiple of 4), K per term (multiple of
                          64 for bf16 / 32 for f32)                                                 */
  int32_t T;           /* frames per sequence: row m -> (b = m / T, t = m % T)                      */
  int32_t groups;      /* independent problems in one launch (>= 1)                                 */
  int32_t epilogue;    /* DN_EPI_*                                                                  */
  const float* bias;   /* [N] fp32 or NULL                                                          */
  int64_t bias_gstride;
  void* out;           /* [M, ldo]                                                                  */
  int32_t ldo;
  int32_t out_dtype;   /* DN_F32 | dtype's 2-byte type | DN_BF16X3 (split rows: ldo, N offsets multiples of 32) */
  int64_t out_gstride;
  const void* res;     /* FILM_GATE: [M, ldr] in res_dtype; RESADD: fp32 [M, ldr] (may alias out)   */
  int32_t ldr;
  int32_t res_dtype;
  int64_t res_gstride;
  const float* gamma_beta; /* FILM_GATE: fp32 [Bc, gb_ld]: gamma at col 0.., beta at col gb_half..; NULL = no FiLM */
  int32_t gb_ld;       /* elements between batch rows (0 = one row shared by the whole batch)       */
  int32_t gb_half;     /* column offset of beta                                                     */
  int64_t gb_gstride;
  const float* pos_table; /* POSEMB: fp32 [T+1, pos_ld] sinusoidal table, row 0 = zeros             */
  int32_t pos_ld;
  int32_t pad_;        /* profiling / tests only.  bit4: 4-column instead of 8-column bf16 stores (the K-loop ablations that
                          used bits 0..3 are compile-time now: -DDN_GEMM_ABL); 0 in every product call; bits 8..15: launch tag
                          (DN_TAG_*) matched by dn_profile_start; bits 16..19: force a tile variant (tests: 1 = 128x128,
                          2 = 256x128, 3 = 256x256, 4 = 256x352 when N % 352 == 0, 6 / 7 = hand-scheduled 256x256 with one / two waves per SIMD [bf16]),
                          0 = chosen from the shape; bits 22 / 23: force the
                          256x352 tile's K order for the taps of a causal conv -- 22 = taps innermost (its default: the
                          activation panel crosses the fabric once, not once per tap), 23 = term-outer (the summation
                          order of every other tile variant)                                                     */
  const int32_t* lengths; /* POSEMB: [B] valid frames per sequence                                  */
  /* RESADD / POSEMB with N <= 512 only: when norm_out != NULL one workgroup owns whole output rows and also emits
   * the NEXT block's RMSNorm of the row it just produced (latent_module.py:620-639, 691, 703):
   * y = out_row / max(|out_row|, 1e-12) * sqrt(norm_D) [* norm_gamma] [* g + b], pad columns zeroed.            */
  void* norm_out;          /* [M, norm_ld] in norm_dtype                                             */
  int32_t norm_ld, norm_dtype, norm_D, norm_gb_ld; /* norm_gb_ld: 0 = one conditioning row for the batch */
  const float* norm_gamma; /* learned gamma [norm_D] or NULL                                         */
  const float* norm_gb;    /* adaptive rows [Bc, norm_gb_ld]: gamma at col 0.., beta at norm_gb_half.. or NULL */
  int32_t norm_gb_half;
  int32_t out_layout;      /* DN_LAYOUT_OUT_KBLOCKED: out is [N/32][M][32] (bf16, GEGLU epilogue, ldo ignored)      */
  /* Split RMSNorm: the norm of a row is divided between the contraction that produces the row and the one that consumes
   * it, so no separate pass over the residual stream remains (x/|x| * sqrt(D) * g + b feeding a Linear W equals
   * (sqrt(D)/|x|) * ((x*g) W^T) + b W^T).
   * Producer (RESADD / POSEMB, N a multiple of 64, any tile): norm_split != 0 makes norm_out receive out_row * gamma
   * (norm_split == 2: in the K-blocked layout [norm_ld/32][M][32], bf16 -- for a consumer that takes DN_LAYOUT_A_KBLOCKED)
   * (norm_gamma, or the gamma half of norm_gb per sample) WITHOUT the 1/|row| factor, and norm_ssq[m, n/64] the sum of
   * squares of the 64 output columns of row m that one wave produced (no atomics: the consumer adds the partials).
   * Consumer (BIAS / SILU / GEGLU): row_ssq != NULL makes out = acc * sqrt(row_D) / max(sqrt(sum_j row_ssq[m, j]), 1e-12)
   * + bias + row_bias[b], with row_bias = beta W^T (+ the layer bias) precomputed by the caller, NULL when there is no beta. */
  int32_t norm_split, norm_ssq_ld;
  float* norm_ssq;         /* [M, norm_ssq_ld]                                                       */
  const float* row_ssq;    /* [M, row_ssq_ld], row_ssq_parts partials per row                        */
  int32_t row_ssq_ld, row_ssq_parts;
  float row_D;             /* the norm's D (not padded)                                              */
  int32_t row_bias_ld;     /* elements between batch rows of row_bias (0 = one row for the batch)    */
  const float* row_bias;   /* [Bc, >= N (packed columns for GEGLU)] fp32 or NULL                     */
  /* GEGLU only (training forward, groups == 1): when pre_out != NULL the epilogue also stores the pre-activation it gates --
   * [M, pre_ld] in out_dtype, packed columns (what the BIAS epilogue would have written) -- for the backward pass.          */
  void* pre_out;
  int32_t pre_ld, pre_pad_;
} DnGemmParams;

int dn_conv_gemm(const DnGemmParams* p, void* stream);
/* 1 when dn_conv_gemm would run this contraction (M, N, K, groups, dtype, epilogue, n_terms are read) on the tile that
 * takes K-blocked A / W terms, else 0.                                                                            */
int dn_conv_gemm_kblocked_ok(const DnGemmParams* p);
/* The tile variant dn_conv_gemm would run this contraction on (1 = 128x128, 2 = 256x128, 3 = 256x256, 4 = 256x352,
 * 5 = whole-row fused norm, 6 / 7 forced-only forms; -1 = K-blocked operands with an unsuitable tile forced): lets a
 * caller lay its buffers out K-blocked only where the contraction lands on a tile that gains from it.              */
int dn_conv_gemm_tile(const DnGemmParams* p);

/* launch tags set by the engine on its dominant contractions */
enum { DN_TAG_FFN_CONV = 1, DN_TAG_WN_DILATED = 2,
       DN_TAG_FFN_CONV_WGRAD = 3 /* training: the weight-gradient contraction of the FFN causal conv */ };

/* Times the next `max_launches` eager dn_conv_gemm launches carrying `tag` with HIP events recorded on their
 * launch stream (not under graph capture); dn_profile_stop synchronises them and returns the average. */
int dn_profile_start(int32_t tag, int32_t max_launches);
int dn_profile_stop(float* avg_ms, int32_t* n_launches);

/* Fused key-masked multi-head self-attention, flash-style (no [B,H,T,T] tensor).
 * Replaces Attend.forward (non-flash branch) latent_module.py:299-343 between the to_q/to_kv and
 * to_out projections (:945-949).  q,k,v,out: row m = b*T+t, head h at columns [h*dh, (h+1)*dh).
 * Keys j >= lengths[b] are masked (lengths[b] == 0 -> uniform over all T keys, as masked_fill gives). */
typedef struct {
  const void* q; const void* k; const void* v; void* out;
  int32_t ldq, ldk, ldv, ldo;
  int32_t B, T, heads, dim_head;
  int32_t dtype;      /* element type of q,k,v,out                                                 */
  int32_t Tk;         /* keys per sequence when they are not the queries' frames (cross-attention, latent_module.py:935-943:
                         k / v row = b*Tk + j); 0 = T (self-attention).  lengths then count valid KEYS (<= Tk)       */
  const int32_t* lengths; /* [B] or NULL (no mask)                                                  */
  float scale;        /* dim_head ** -0.5                                                           */
  int32_t pad2_;
  float* lse;         /* optional fp32 [B, heads, T]: log2-domain log-sum-exp of the scaled scores of every query
                         (max * scale * log2(e) + log2(sum)), what dn_attention_backward recomputes P from; NULL = not kept */
  float dropout_p;    /* training only (latent_module.py:338,668: nn.Dropout(0.1) on the attention probabilities): probability of
                         zeroing a probability; kept ones are scaled by 1 / (1 - p) AFTER the softmax normalisation.  0 = off.
                         The keep mask is a counter-based hash of (seed, batch, head, query, key) -- dn_attention_backward
                         regenerates it from the same seed; tests/test_hip_train_ops.py restates the hash on the host         */
  uint32_t seed_lo, seed_hi;
  int32_t pad3_;
} DnAttnParams;

int dn_attention(const DnAttnParams* p, void* stream);

/* RMSNorm (latent_module.py:620-639): y = x / max(|x|_2, 1e-12) * sqrt(D) [* gamma] [* g_c[b] + b_c[b]].
 * x fp32 [M, ldx]; y [M, ldy] in out_dtype; gamma fp32 [D] or NULL; gamma_beta fp32 [Bc, gb_ld] or NULL. */
int dn_rmsnorm(const float* x, int32_t ldx, void* y, int32_t ldy, int32_t out_dtype, int32_t M, int32_t D,
               int32_t T, const float* gamma, const float* gamma_beta, int32_t gb_ld, int32_t gb_half,
               void* stream);

/* LearnedSinusoidalPosEmb + Linear + SiLU (latent_module.py:104-116, 741-745).
 * times int32 [B]; w_freq fp32 [half]; W fp32 [C, 2*half+1]; bias fp32 [C]; out fp32 [B, ldo] and,
 * when out_act != NULL, a copy in act_dtype [B, ldo] for the conditioning contractions.            */
int dn_time_cond(const int32_t* times, int32_t B, const float* w_freq, int32_t half, const float* W,
                 const float* bias, int32_t C, float* out, void* out_act, int32_t act_dtype, int32_t ldo,
                 void* stream);

/* DDIM eta=0 update (latent_module.py:1419-1442, safe_div :958-959), elementwise over [B, T*z... rows].
 * coef fp32 [n_steps, 4] = {sqrt_abar, sqrt_1m_abar, sqrt(abar_prev), sqrt(1-abar_prev)} (fp32 casts
 * formed as the reference does); t int32 [B].  x, eps, x_out fp32 [M, ld]; x_act (optional, [M, ld_act])
 * receives x_out in act_dtype for the next step's first contraction.                                        */
int dn_ddim_step(const float* x, const float* eps, float* x_out, void* x_act, int32_t act_dtype,
                 int32_t ld_act, int32_t M, int32_t C, int32_t ld, int32_t T, const float* coef,
                 const int32_t* t, void* stream);

/* out = a[t_b]*x + b[t_b]*noise: q_sample / noise injection
 * (latent_module.py:1405-1409, 1538-1543; diffusion/gaussian_diffusion.py:215-230).               */
int dn_q_sample(const float* x, const float* noise, float* out, void* out_act, int32_t act_dtype,
                int32_t ld_act, int32_t M, int32_t C, int32_t ld, int32_t T, const float* coef_a,
                const float* coef_b, const int32_t* t, void* stream);

/* One reverse step of GaussianDiffusion for an eps-predicting model (diffusion/gaussian_diffusion.py:254-417 p_mean_variance
 * + p_sample, :513-560 ddim_sample): pred_xstart = sqrt_recip*x - sqrt_recipm1*eps [clamped to +-1], posterior mean,
 * FIXED_LARGE / FIXED_SMALL (table column 4) or LEARNED_RANGE variance (model_out has 2x channels on dim 1), then
 * sampler 0: mean + 1[t!=0]*exp(.5 logvar)*noise, sampler 1: DDIM with `eta`.  Tensors are fp32 [N, inner] (inner = C*L
 * contiguous; model_out [N, 2*inner] when learned_range).  table: fp32 [T, DN_GD_COLS].                              */
#define DN_GD_COLS 12
typedef struct {
  const float* x; const float* model_out; const float* noise; /* noise may be NULL (treated as 0) */
  float* sample; float* pred_xstart;                          /* pred_xstart may be NULL */
  const int32_t* t;                                           /* [N] */
  const float* table;
  int32_t N, inner, learned_range, clip_denoised, sampler;
  float eta;
  int32_t predict_xstart;   /* ModelMeanType.START_X (:317-318): model_out IS the x_0 prediction (clamped when clip_denoised) */
  const float* cond_grad;   /* optional fp32 [N, inner]: cond_fn(x, t), the gradient of a conditional log-probability -- sampler 0:
                               condition_mean (:346-358) mean += variance * grad; sampler 1: condition_score (:360-374)
                               eps -= sqrt(1 - abar) * grad, pred_xstart and the mean re-derived from it                        */
} DnGaussianStep;
int dn_gaussian_step(const DnGaussianStep* p, void* stream);

/* The moments and loss terms of GaussianDiffusion for an eps-predicting model, elementwise over fp32 [N, inner] tensors (every
 * output pointer optional):
 *   model_out != NULL: p_mean_variance (diffusion/gaussian_diffusion.py:254-332) -> mean, variance, log_variance, pred_xstart
 *     (_predict_xstart_from_eps :334-339, clipped to +-1 when clip_denoised); reverse_sample = the DDIM reverse-ODE step
 *     (ddim_reverse_sample :562-598); with x_start also vb = the per-element term of _vb_terms_bpd (:682-713) in bits:
 *     KL(q(x_{t-1}|x_t,x_0) || p) for t > 0, the discretised-Gaussian decoder NLL (diffusion_utils.py:66-88) at t = 0.
 *   model_out == NULL, x_start != NULL: q_posterior_mean_variance (:232-252) -> mean, variance, log_variance.
 * Table columns (fp32 casts of the float64 schedule): 0 sqrt_recip_abar, 1 sqrt_recipm1_abar, 2 post_coef1, 3 post_coef2,
 * 4 fixed log variance, 5 posterior_log_variance_clipped (= min_log), 6 log beta (= max_log), 7 abar, 8 abar_prev, 9 abar_next,
 * 10 posterior_variance, 11 fixed variance.                                                                              */
typedef struct {
  const float* x; const float* model_out; const float* x_start;
  const int32_t* t; const float* table;
  float *mean, *variance, *log_variance, *pred_xstart, *vb, *reverse_sample;
  int32_t N, inner, learned_range, clip_denoised;
  int32_t predict_xstart;   /* ModelMeanType.START_X: model_out is the x_0 prediction */
} DnGaussianMoments;
int dn_gaussian_moments(const DnGaussianMoments* p, void* stream);

/* DiagonalGaussianDistribution (distributions.py:24-41, 62-74): params fp32 [M, ldp] = [mean ; logvar];
 * z = mean + exp(0.5*clamp(logvar,-30,20))*noise; kl_rows (optional, fp32 [M]) = 0.5*sum_c(mean^2+var-1-logvar)
 * for valid frames, 0 for pads.                                                                    */
int dn_posterior_sample(const float* params, int32_t ldp, const float* noise, int32_t ldn, float* z,
                        void* z_act, int32_t act_dtype, int32_t ldz, int32_t M, int32_t Z, int32_t T,
                        const int32_t* lengths, float* kl_rows, void* stream);

/* argmax over the first V logits of each row minus `offset` (latent_module.py:1450-1451). */
int dn_argmax_units(const float* logits, int32_t ld, int32_t M, int32_t V, int32_t offset, int32_t* units,
                    void* stream);

/* Philox4x32-10 standard normal fill (Box-Muller), the build's own generator for throughput runs
 * (the reference draws with torch.randn, latent_module.py:1409, distributions.py:38).             */
int dn_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);

/* dtype conversion / zero-padded row copy: dst[m, 0:C] = src[m, 0:C], dst[m, C:ldd] = 0. */
int dn_convert_rows(const void* src, int32_t src_dtype, int32_t lds, void* dst, int32_t dst_dtype,
                    int32_t ldd, int32_t M, int32_t C, void* stream);

/* ------------------------------------------------------------------ optimizer step */
/* The update of the reference's training recipe (scripts/diffusion/train.sh:29-31: Adam, betas (0.9, 0.98),
 * --clip-norm 2.0, inverse_sqrt schedule) over flat fp32 buffers; the schedule itself is host arithmetic
 * (diffnorm_amd/optim.py).  Backward kernels are not part of this round.                                      */

/* sumsq[0] (+)= sum_i grad[i]^2: the square of fairseq.utils.clip_grad_norm_'s total_norm (fairseq/utils.py:347-390;
 * call once per gradient buffer with accumulate != 0 after the first).  scratch: 1024 floats.  Two launches, fixed
 * summation order (bit-reproducible).                                                                          */
int dn_grad_sumsq(const float* grad, int64_t n, float* scratch, float* sumsq, int32_t accumulate, void* stream);

typedef struct {
  double lr, beta1, beta2, eps, weight_decay; /* doubles, as the reference's Python floats: 1 - beta and the step size
                                                 are formed in double and rounded to fp32 once                  */
  double max_norm;  /* --clip-norm; <= 0 or sumsq == NULL: no clipping                                          */
  int32_t step;     /* 1 for the first update (fairseq/optim/adam.py:212)                                       */
  int32_t pad_;
  double grad_scale; /* the trainer's multiply_grads (fairseq/trainer.py:918-933: world / sample_size after DDP's mean, i.e.
                        1 / sample_size on a summed gradient): applied to every gradient before the norm and the update;
                        0 is read as 1                                                                            */
  const float* grad_scale_dev; /* optional device scalar multiplied into grad_scale (a sample size that only exists on the
                                  device after the statistics all-reduce), NULL = 1                                */
} DnAdamParams;

/* Operand preparation for the weight gradient of a causal conv / Linear (the contraction over frames): bf16 [B*T, ld]
 * frames-major -> channels-major with `front` zero frames in front of every sequence and zeros up to Tp frames and in rows
 * >= C; the B*Tp columns are stored as [B*Tp / chunk][rows_total][chunk] (chunk a multiple of 64), one K-slice per group
 * of a split-K dn_conv_gemm; the call fills rows [row0, row0 + rows) (the taps of a conv stack their operands).  dW_j = dY^T . shift_j(X) is then that contraction with A = transposed dY (front 0), W =
 * transposed X with front = shift_j (the padding supplies the zeros of the causal left context), groups = K-slices spread
 * over the chip, fp32 partial outputs summed by the caller (diffnorm_amd/ops.py: conv_weight_grad).                    */
int dn_transpose_pad(const void* src, int32_t ld, int32_t B, int32_t T, int32_t C, int32_t front, int32_t Tp, void* dst,
                     int32_t rows, int32_t rows_total, int32_t row0, int32_t chunk, void* stream);

/* One fairseq Adam update (fairseq/optim/adam.py:159-239) of n fp32 parameters in place, with the gradient scaled by
 * s = grad_scale and then by min(1, max_norm / (s * sqrt(sumsq[0]) + 1e-6)) first (fairseq/trainer.py:918-933, fairseq/utils.py:
 * 392-396; sumsq is the sum of squares of the UNSCALED buffer, which is itself left untouched); param_bf16 != NULL also receives the updated parameters in bf16 (the forward kernels' operand type). */
int dn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const DnAdamParams* hp,
                 const float* sumsq, void* param_bf16, void* stream);

/* ------------------------------------------------------------------ backward ops ---------- */
/* The contractions of a backward pass are dn_conv_gemm itself (data gradient: negative shifts + transposed packed weights from
 * dn_transpose_weights; weight gradient: dn_transpose_pad operands + split-K groups + dn_wgrad_reduce).  The entry points
 * below are the backward of everything that is not a contraction, the loss gradients and the reductions over frames.  All
 * reductions are two-stage with fixed block counts (no atomics): a training step is bit-reproducible.                     */

/* Gradient of dn_attention (Attend.forward non-flash branch, latent_module.py:299-343): P is recomputed from the forward's
 * per-query log-sum-exp (DnAttnParams.lse); dq / dk / dv have the layout of q / k / v (row m = b*T+t, head h at columns
 * [h*dh, (h+1)*dh)) with their own row strides, so they can be three column blocks of one [M, 3*heads*dh] buffer.
 * delta: fp32 scratch [B, heads, T] (sum_d dO*O per query, written by the call).  No atomics: dk/dv and dq each have one
 * writer.  Attention dropout (p = 0.1 in the reference's training mode, :338,668): dropout_p / seed as in the forward.        */
typedef struct {
  const void *q, *k, *v, *out, *dout;
  void *dq, *dk, *dv;
  int32_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  int32_t B, T, heads, dim_head;
  int32_t dtype;
  int32_t pad_;
  const int32_t* lengths; /* [B] or NULL */
  float scale;
  int32_t pad2_;
  const float* lse;       /* [B, heads, T] from the forward */
  float* delta;           /* [B, heads, T] scratch */
  float dropout_p;        /* must equal the forward's (with the same seed): the mask is regenerated, not stored */
  uint32_t seed_lo, seed_hi;
  int32_t pad3_;
} DnAttnBwdParams;
int dn_attention_backward(const DnAttnBwdParams* p, void* stream);

/* Gradient of dn_time_cond (latent_module.py:104-116, 741-745): dcond fp32 [B, ldd] -> ACCUMULATES dW [C, 2*half+1], dbias [C] and
 * dw_freq [half] (the learned Fourier frequencies: d/dw sin(2 pi w t) = 2 pi t cos(.)).  ds_scratch: fp32 [B, C].            */
int dn_time_cond_backward(const int32_t* times, int32_t B, const float* w_freq, int32_t half, const float* W, const float* bias, int32_t C,
                          const float* dcond, int32_t ldd, float* ds_scratch, float* dw_freq, float* dW, float* dbias, void* stream);

/* WaveNet gate as a stand-alone pass (the training forward keeps the pre-activation h that its derivative needs):
 * out = tanh(h') sigmoid(h') + res with h' = h * gamma[b] + beta[b] when gamma_beta != NULL (latent_module.py:525-530).
 * h, res, out, dout, dh: [M, ld] in `dtype`, same ld.  Backward: dh = dout * gate'(h') [* gamma]; with FiLM, dgb_rows
 * (fp32 [M, dgb_ld], optional) receives the per-frame terms of the conditioning gradient: [dg * h | dg] at [c | gb_half + c]. */
int dn_gate_forward(const void* h, const void* res, void* out, int32_t dtype, int32_t M, int32_t ld, int32_t T,
                    const float* gamma_beta, int32_t gb_ld, int32_t gb_half, void* stream);
int dn_gate_backward(const void* dout, const void* h, void* dh, int32_t dtype, int32_t M, int32_t ld, int32_t T,
                     const float* gamma_beta, int32_t gb_ld, int32_t gb_half, float* dgb_rows, int32_t dgb_ld, void* stream);

/* GEGLU (latent_module.py:881-884) on the packed column order of the GEGLU projection (DN_EPI_GEGLU: per 16 columns, 8 values
 * then the 8 gates of the same outputs): pre [M, 2*ip] -> out [M, ip] = gelu_erf(gate) * value; backward fills dpre [M, 2*ip]. */
int dn_geglu_forward(const void* pre, void* out, int32_t dtype, int32_t M, int32_t ip, void* stream);
int dn_geglu_backward(const void* dout, const void* pre, void* dpre, int32_t dtype, int32_t M, int32_t ip, void* stream);

/* Gradient of dn_rmsnorm (latent_module.py:620-639).  x fp32 [B*T, ldx] (the norm's input), dy [B*T, lddy] in dy_dtype;
 * dx fp32 [B*T, ldx] = norm gradient (+ dres, the gradient arriving on the residual branch, when not NULL; dx may alias dres);
 * dx_act (optional): dx once more in act_dtype [B*T, ld_act], pad columns zero (the next contraction's operand).
 * Parameter gradients are ACCUMULATED: dgamma fp32 [D] (learned gamma) or dgamma_beta fp32 [B, dgb_ld] = [d g_c | d b_c] at
 * [c | gb_half + c] per sample (adaptive norm).  scratch: dn_rmsnorm_backward_scratch_bytes(B, T, D).  D <= 1024.          */
size_t dn_rmsnorm_backward_scratch_bytes(int32_t B, int32_t T, int32_t D);
int dn_rmsnorm_backward(const float* x, int32_t ldx, const void* dy, int32_t lddy, int32_t dy_dtype, int32_t B, int32_t T, int32_t D,
                        const float* gamma, const float* gamma_beta, int32_t gb_ld, int32_t gb_half, const float* dres, float* dx,
                        void* dx_act, int32_t act_dtype, int32_t ld_act, float* dgamma, float* dgamma_beta, int32_t dgb_ld,
                        float* scratch, void* stream);

/* out[g, c] (+)= scale * sum over the rows of group g of src[row, c] (bias gradients, per-sample conditioning gradients,
 * loss sums): src [groups * rows_per_group, ld] in dtype, out fp32 [groups, out_ld].  scratch: dn_colsum_scratch_bytes.    */
size_t dn_colsum_scratch_bytes(int32_t groups, int32_t rows_per_group, int32_t C);
int dn_colsum(const void* src, int32_t ld, int32_t dtype, int32_t groups, int32_t rows_per_group, int32_t C, float* out,
              int32_t out_ld, float scale, int32_t accumulate, float* scratch, void* stream);

/* Gradient of dn_posterior_sample + the KL term (distributions.py:24-41, 62-74): dparams [M, ldo] (act_dtype, pad columns
 * zero) = [d mean ; d logvar] from dz fp32 [M, lddz] and kl_weight = (loss weight of the KL) / (B * Z * T); the clamp of
 * logvar to [-30, 20] passes no gradient outside the range, KL only counts valid frames.                                 */
int dn_posterior_backward(const float* params, int32_t ldp, const float* noise, int32_t ldn, const float* dz, int32_t lddz,
                          void* dparams, int32_t act_dtype, int32_t ldo, int32_t M, int32_t Z, int32_t T, const int32_t* lengths,
                          float kl_weight, void* stream);

/* Label-smoothed cross entropy over log_softmax(logits) and its gradient (fairseq/criterions/label_smoothed_cross_entropy.py:
 * 34-51 with ignore_index 0; speech_vae_decoder_loss.py:60-79): rows fp32 [M, 4] = {nll, smooth, correct, valid} per frame
 * (nll = -lprob[target], smooth = -sum lprobs; loss = (1 - eps - eps_i) nll + eps_i smooth, eps_i = eps / (V - 1));
 * dlogits (optional, act_dtype [M, ldd], pad columns zero) = grad_scale * d loss / d logits.  V <= 1024.                  */
int dn_lsce_loss_grad(const float* logits, int32_t ld, const int32_t* target, int32_t M, int32_t V, float epsilon, float grad_scale,
                      float* rows, void* dlogits, int32_t act_dtype, int32_t ldd, void* stream);

/* Masked MSE (latent_module.py:1135-1138, 1576-1577: mean over the elements of valid frames): sq_rows fp32 [M] = sum_c
 * (pred - target)^2 of valid frames (0 for pads); dpred fp32 [M, ldd] (+)= grad_scale * (pred - target), and dpred_act its
 * copy in act_dtype [M, ld_act] (both optional, pad columns and pad frames zero).                                          */
int dn_masked_mse_grad(const float* pred, int32_t ldp, const float* target, int32_t ldt, int32_t M, int32_t C, int32_t T,
                       const int32_t* lengths, float grad_scale, float* sq_rows, float* dpred, int32_t ldd, int32_t accumulate,
                       void* dpred_act, int32_t act_dtype, int32_t ld_act, void* stream);

/* out[0] (+)= sum_i v[i] (fixed two-stage order); scratch: 256 floats. */
int dn_vec_sum(const float* v, int64_t n, float* out, int32_t accumulate, float* scratch, void* stream);

/* dst[i] = sum_{k < count} src[k * stride + i], i < n (the gradients a tensor receives from several consumers). */
int dn_sum_groups(const void* src, int64_t stride, int32_t count, void* dst, int32_t dtype, int64_t n, void* stream);
/* out[b][c] = sum_n X[b][n] * W[n][c], fp32: a handful of rows (one stream of W per 32) against a huge ROW-MAJOR matrix W [N, ldw] (C valid columns), streamed once
 * as it lies -- the data gradient of the eps-predictor's conditioning projection (autograd of nn.Linear, latent_module.py:841-852),
 * without a transposed copy of its 470 MB.  Partial sums per row slice in `scratch` (dn_rows_times_weight_scratch_bytes), added in a
 * fixed order.                                                                                                                     */
size_t dn_rows_times_weight_scratch_bytes(int32_t B, int32_t N, int32_t C);
int dn_rows_times_weight(const float* X, int32_t ldx, int32_t B, const float* W, int32_t ldw, int32_t N, int32_t C, float* out,
                         float* scratch, void* stream);

/* Batched padded transpose of packed weights, the W operand of the data-gradient contraction: dst[n] [Cp][Rp] = (the first
 * R rows of src[n], [R][Cc])^T, zeros beyond; matrices src_stride / dst_stride elements apart.                           */
int dn_transpose_weights(const void* src, int32_t dtype, int32_t count, int64_t src_stride, int32_t R, int32_t Cc, void* dst,
                         int64_t dst_stride, int32_t Rp, int32_t Cp, void* stream);

/* Last step of a weight gradient: grad[tap][n][k] += sum_s part[s][n][tap * rows_w + k] for n < cout, k < Kp; part is the fp32
 * split-K output [slices][cout][n_total] of the contraction over frames, grad the packed fp32 layout [n_taps][Np][Kp].    */
int dn_wgrad_reduce(const float* part, int32_t slices, int32_t cout, int32_t n_total, int32_t rows_w, int32_t n_taps, float* grad,
                    int32_t Np, int32_t Kp, void* stream);

/* dst[j * ld + c] += src[c], j < count (the L skip-conv biases of a WaveNet share one gradient). */
int dn_add_broadcast(const float* src, float* dst, int32_t C, int32_t ld, int32_t count, void* stream);

/* The same weight gradient straight from the row-major operands, no transposed copies (bf16; autograd of CausalConv1d
 * latent_module.py:476-485 / nn.Linear): sum over frames m of dY[m][n] * X_tap[m - shift_tap][k] (zero where the frame index of m
 * within its sequence is < shift_tap), the contraction over frames fed to the MFMA through transposing LDS reads
 * (csrc/wgrad_tn.hip).  dY [B*T][lddy], X_tap [B*T][ldx[tap]] bf16, rows 16-byte multiples (pad columns may hold anything).
 * slices == 1: grad[tap][padn(cout)][padk(cin)] += result (part == NULL); slices > 1:
 * part[slice][cout][n_taps * padn(cin)] = the partial sums of that slice of the frames (then dn_wgrad_reduce), grad unused.  */
int dn_conv_weight_grad_tn(const void* dy, int32_t lddy, int32_t cout, const void* const* x, const int32_t* ldx, const int32_t* shift,
                           int32_t n_taps, int32_t cin, int32_t B, int32_t T, int32_t slices, float* part, float* grad, void* stream);

/* dn_transpose_pad for fp32 operands (exact-fp32 mode); chunk a multiple of 32. */
int dn_transpose_pad_f32(const float* src, int32_t ld, int32_t B, int32_t T, int32_t C, int32_t front, int32_t Tp, float* dst,
                         int32_t rows, int32_t rows_total, int32_t row0, int32_t chunk, void* stream);

/* ------------------------------------------------------------------ whole-path engine ---------- */

typedef struct {
  int32_t dim, latent, depth, heads, dim_head, wn_layers, wn_stacks, cond_mult;
  int32_t dtype;  /* DN_F32 | DN_BF16 */
  int32_t max_pos; /* rows in the positional table minus 1 */
  /* conditional variant (use_cond=True, latent_module.py:752-773): 0 = the unconditional model of the recipe */
  int32_t dim_prompt, num_latents, resampler_depth;
} DnEpsConfig;

typedef struct {
  int32_t dim, z, depth, heads, dim_head, stacks, layers, vocab;
  int32_t n_mults; int32_t mults[4];
  int32_t dtype;
} DnVaeConfig;

typedef struct DnEps DnEps;   /* eps-predictor `Model`        latent_module.py:709-876  */
typedef struct DnVae DnVae;   /* SpeechVAEEncoderDecoder      latent_module.py:1035-1142 */

/* Packed-weight tables: device pointers in the order produced by diffnorm_amd/packing.py
 * (documented in DESIGN.md "packed layout"); the library keeps the pointers, not copies.           */
int dn_eps_create(const DnEpsConfig* cfg, const void* const* weights, int32_t n_weights, DnEps** out);
void dn_eps_destroy(DnEps* m);
size_t dn_eps_workspace_bytes(const DnEps* m, int32_t B, int32_t T);
/* Model.forward (latent_module.py:828-876): x fp32 [B,T,latent] dense, t int32 [B], lengths int32 [B]
 * -> eps fp32 [B,T,latent].  shared_t != 0 promises all t[b] equal (sampling) so the 56 conditioning
 * projections run for one row.                                                                     */
int dn_eps_forward(DnEps* m, const float* x, const int32_t* t, const int32_t* lengths, int32_t B, int32_t T,
                   int32_t shared_t, float* eps_out, void* workspace, size_t workspace_bytes, void* stream);

/* Conditional variant (Model.forward with condition_on_prompt, latent_module.py:828-876, PerceiverResampler :416-471,
 * classifier-free guidance :813-826): prompt fp32 [B, Tp, dim_prompt] with prompt_lengths [B]; drop int32 [B] = the guidance drop
 * mask (1: the sample runs on null_prompt_cond / null_prompt_tokens).  The pooled-prompt condition is concatenated to the time
 * condition (2x conditioning width), the resampled prompt latents feed a cross-attention block in every transformer layer.
 * Needs a model created with cfg.dim_prompt > 0 (packed table: diffnorm_amd/packing.py::pack_eps).                        */
size_t dn_eps_cond_workspace_bytes(const DnEps* m, int32_t B, int32_t T, int32_t Tp);
int dn_eps_forward_cond(DnEps* m, const float* x, const int32_t* t, const int32_t* lengths, const float* prompt,
                        const int32_t* prompt_lengths, const int32_t* drop, int32_t B, int32_t T, int32_t Tp, float* eps_out,
                        void* workspace, size_t workspace_bytes, void* stream);
/* The same pass inside a CHAIN of steps (the prompted, guided sampling loop): everything that depends on the prompt only -- the
 * pooled-prompt half of the conditioning projection, the PerceiverResampler and every layer's cross-attention keys / values -- is
 * kept in the workspace, and with DN_COND_REUSE_PROMPT a later call on the SAME workspace and shapes skips it (the caller
 * guarantees nothing else wrote the workspace in between and that prompt / prompt_lengths / drop are unchanged).  time_table
 * (optional, fp32 [table_n, n_cond] from dn_eps_cond_time_table, first row = timestep table_t0): the time half of the conditioning
 * rows of every step of the chain, so the 2 C x n_cond projection is not streamed inside the loop.  flags = 0, time_table = NULL
 * is dn_eps_forward_cond.                                                                                                     */
#define DN_COND_REUSE_PROMPT 1
int dn_eps_forward_cond_ex(DnEps* m, const float* x, const int32_t* t, const int32_t* lengths, const float* prompt,
                           const int32_t* prompt_lengths, const int32_t* drop, int32_t B, int32_t T, int32_t Tp, float* eps_out,
                           void* workspace, size_t workspace_bytes, int32_t flags, const float* time_table, int32_t table_t0,
                           int32_t table_n, void* stream);
size_t dn_eps_cond_time_table_workspace_bytes(const DnEps* m, int32_t n_t);
int dn_eps_cond_time_table(DnEps* m, int32_t t0, int32_t n_t, float* table, void* workspace, size_t workspace_bytes, void* stream);

int dn_vae_create(const DnVaeConfig* cfg, const void* const* weights, int32_t n_weights, DnVae** out);
void dn_vae_destroy(DnVae* m);
size_t dn_vae_workspace_bytes(const DnVae* m, int32_t B, int32_t T);
/* encoder WaveNets of encode_feature (latent_module.py:1099-1106): feat fp32 [B,T,dim] -> params fp32 [B,T,2z] */
int dn_vae_encode_params(DnVae* m, const float* feat, int32_t B, int32_t T, float* params, void* workspace,
                         size_t workspace_bytes, void* stream);
/* decode_feature (latent_module.py:1109-1116): latent fp32 [B,T,z] -> recon fp32 [B,T,dim], logits fp32
 * [B,T,vocab] (either may be NULL), units int32 [B,T] = argmax-4 (may be NULL)                     */
int dn_vae_decode(DnVae* m, const float* latent, const int32_t* lengths, int32_t B, int32_t T, float* recon,
                  float* logits, int32_t* units, void* workspace, size_t workspace_bytes, void* stream);

/* LatentDiscreteModel.ddim_sample's device loop (latent_module.py:1405-1445): starting from x (already
 * noised at start_step), evaluates the eps-predictor for t = start_step-1 .. 1 (t = 0 only when
 * start_step == 1) with the eta=0 update after each.  coef: fp32 [timesteps,4] as dn_ddim_step.
 * x fp32 [B,T,latent] is updated in place.  The conditioning rows of all steps are built once, in
 * fp32, before the loop.  max_evals > 0 stops after that many evaluations (the caller continues with
 * start_step - max_evals).  flags: DN_LOOP_* bits.  Workspace:
 * dn_ddim_workspace_bytes.  Returns the number of model evaluations (>= 0) or a negative error.                 */
enum { DN_LOOP_GRAPH = 1,  /* capture one step into a hipGraph and replay it */
       DN_LOOP_SPLIT2 = 2, /* run the two half-batches as parallel branches (side stream / forked graph) */
       DN_LOOP_KEEP_TABLE = 4 /* continuing a chain (after a max_evals stop): the caller guarantees that nothing wrote the
                                 workspace since the previous dn_ddim_loop call with the same B, T and split; the
                                 conditioning table built then (rows t < its start_step) is reused instead of rebuilt */ };
size_t dn_ddim_workspace_bytes(const DnEps* m, int32_t B, int32_t T, int32_t start_step);
int dn_ddim_loop(DnEps* m, float* x, const int32_t* lengths, int32_t B, int32_t T, int32_t start_step,
                 int32_t max_evals, const float* coef, int32_t timesteps, int32_t flags, void* workspace,
                 size_t workspace_bytes, void* stream);

/* The same device loop with the ancestral (DDPM) update of GaussianDiffusion.p_sample / p_sample_loop
 * (diffusion/gaussian_diffusion.py:376-417, 459-511) instead of DDIM eta = 0 -- BASELINE configs[2] read literally: for
 * t = start_step-1 .. 0 (t = 0 IS evaluated, its noise masked): eps = Model(x, t); x0 = sqrt_recip_abar x - sqrt_recipm1_abar eps
 * [clamped to +-1 when clip_denoised]; mean = coef1 x0 + coef2 x; x <- mean + 1[t != 0] exp(0.5 log_var_t) z.
 * table: fp32 [timesteps, DN_GD_COLS] as dn_gaussian_step takes it (column 4 = the fixed log-variance: FIXED_SMALL or
 * FIXED_LARGE).  z: injected -- noise fp32 [start_step, B*T*latent], row (start_step-1-t) used at step t (parity runs) -- or,
 * noise == NULL, drawn in the update kernel from Philox4x32-10 keyed by `seed` with the counter (t, element quad): the step
 * index is read from the loop's device counter, so a captured graph draws fresh noise at every replay.  Everything else
 * (conditioning table once per chain, hipGraph replay, two half-batch streams, max_evals / KEEP_TABLE) as dn_ddim_loop;
 * workspace: dn_ddim_workspace_bytes.  Returns the number of model evaluations or a negative error.                     */
int dn_ddpm_loop(DnEps* m, float* x, const int32_t* lengths, int32_t B, int32_t T, int32_t start_step, int32_t max_evals,
                 const float* table, int32_t timesteps, int32_t clip_denoised, uint64_t seed, const float* noise, int32_t flags,
                 void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ VAE training step (BASELINE config 4) */
/* speech_vae_decoder_loss training (reference SpeechVAEEncoderDecoder.forward latent_module.py:1118-1142 + the criterion
 * fairseq/criterions/speech_vae_decoder_loss.py:45-95) on flat buffers in the PACKED parameter layout (csrc/engine.h: rows
 * padded to 128, K to 64, GEGLU interleave, conv taps as matrices; transformer tensors layer-major; zero pads stay zero under
 * Adam without weight decay):
 *   master fp32 [n]  what dn_adam_step updates          work [n] in cfg.dtype  (bf16: dn_adam_step's bf16 copy; f32: == master)
 *   grads  fp32 [n]  dn_vae_train_backward ADDS into it  aux  bytes            transposed matrices + summed skip biases,
 *                                                                              rebuilt by dn_vae_train_refresh after an update
 * diffnorm_amd/packing.py::pack_vae_train converts a state dict in the reference layout to / from this layout. */
typedef struct DnVaeTrain DnVaeTrain;
typedef struct {
  const float* feat;        /* [B, T, dim] fp32 target features (also the encoder input)                               */
  const int32_t* units;     /* [B, T] dictionary indices (unit + 4), 0 = pad                                           */
  const int32_t* lengths;   /* [B]                                                                                     */
  const float* noise;       /* [B, T, z] posterior noise (distributions.py:38-40 draws it on the CPU generator)        */
  int32_t B, T;
  int32_t ntokens;          /* sample["ntokens"] = sum of lengths                                                      */
  float w_lsce, w_mse, w_kl; /* criterion weights: 0.1, 10, 1e-4 (speech_vae_decoder_loss.py:80-83)                     */
  float label_smoothing;    /* 0.1                                                                                     */
  float loss_scale;         /* multiplies every gradient (1 normally)                                                   */
  float* stats;             /* fp32 [8] out: loss, nll_loss, mse_loss, kl_loss, acc, n_valid, lsce/ntokens, 0            */
  float* logits_out;        /* optional fp32 [B, T, vocab]                                                             */
  float* recon_out;         /* optional fp32 [B, T, dim]                                                               */
  const float* ext_dlogits; /* backward only, optional fp32 [B, T, vocab]: d loss / d logits supplied by the caller (a criterion
                               that differentiates the logits itself) instead of the fused LS-CE gradient; w_mse / w_kl are
                               then d loss / d mse_loss and d loss / d kl_loss                                          */
  float attn_dropout;       /* dropout on the attention probabilities (latent_module.py:338,668: 0.1 in train mode, 0 in eval) */
  uint32_t dropout_seed_lo, dropout_seed_hi; /* the mask is a counter hash of (seed, layer, batch, head, query, key): forward and
                               backward of one step must be given the same seed; change it every update                    */
  int32_t pad_;
} DnVaeTrainBatch;

int dn_vae_train_create(const DnVaeConfig* cfg, DnVaeTrain** out);
void dn_vae_train_destroy(DnVaeTrain* m);
int64_t dn_vae_train_param_count(const DnVaeTrain* m);   /* n: elements of master / work / grads                       */
size_t dn_vae_train_aux_bytes(const DnVaeTrain* m);
/* element offsets of the packed tensors in table order (2 * n_wave * 10 + 10 * depth + 4 entries); returns the count   */
int dn_vae_train_offsets(const DnVaeTrain* m, int64_t* offsets, int32_t capacity);
/* [offset, offset + count) of the gradient buffer that backward stage `stage` completes                               */
int dn_vae_train_stage_range(const DnVaeTrain* m, int32_t stage, int64_t* offset, int64_t* count);
int dn_vae_train_bind(DnVaeTrain* m, float* master, void* work, void* aux, float* grads);  /* 256-byte aligned buffers   */
int dn_vae_train_refresh(DnVaeTrain* m, void* stream);
size_t dn_vae_train_workspace_bytes(const DnVaeTrain* m, int32_t B, int32_t T);
/* forward + losses; every activation the backward needs stays in the workspace                                          */
int dn_vae_train_forward(DnVaeTrain* m, const DnVaeTrainBatch* batch, void* workspace, size_t workspace_bytes, void* stream);
/* backward stages first_stage .. last_stage of the step dn_vae_train_forward just ran on this workspace: 0 = loss gradients +
 * decoder_lm + to_pred, 1 .. depth = transformer layers depth-1 .. 0, depth+1 = decoder WaveNets + posterior, depth+2 =
 * encoder WaveNets.  Gradient ranges complete in reverse parameter order (dn_vae_train_stage_range), so the caller can start
 * the all-reduce of a finished range while later stages still run.                                                       */
int dn_vae_train_backward(DnVaeTrain* m, const DnVaeTrainBatch* batch, int32_t first_stage, int32_t last_stage, void* workspace,
                          size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ mask-predict update */
/* The per-iteration update of the CMLM decoder downstream of the normalised units (fairseq/models/nat/cmlm_transformer.py:88-134,
 * called from the loop of fairseq/iterative_refinement_generator.py:200-230): positions holding `unk` (the mask symbol) take
 * argmax / max of log_softmax(logits); `predicted` receives the tokens at that point; then, unless step + 1 == max_step, the
 * trunc((n_nonpad - 2) * (1 - (step + 1) / max_step)) lowest-scoring positions are re-masked (_skeptical_unmasking :19-25; equal
 * scores ordered by position).  logits fp32 [B, T, V]; tokens int32 [B, T] and scores fp32 [B, T] updated in place.  T <= 2048.  */
int dn_cmlm_step(const float* logits, int32_t* tokens, float* scores, int32_t* predicted, int32_t B, int32_t T, int32_t V, int32_t step,
                 int32_t max_step, int32_t unk, int32_t pad, void* stream);
/* The same update with the iteration index read from the DEVICE (int32 *step_dev): one refinement iteration of the loop of
 * fairseq/iterative_refinement_generator.py:200-230 -- dn_nar_decoder_forward + this -- can then be captured into a hipGraph once and
 * replayed while a device counter advances (diffnorm_amd/nar_decoder.py).                                                         */
int dn_cmlm_step_dev(const float* logits, int32_t* tokens, float* scores, int32_t* predicted, int32_t B, int32_t T, int32_t V,
                     const int32_t* step_dev, int32_t max_step, int32_t unk, int32_t pad, void* stream);

/* ------------------------------------------------------------------ diffusion training step */
/* LatentDiscreteModel.forward (latent_module.py:1514-1613; criterion fairseq/criterions/ddpm_discrete_loss.py:37-75) for the
 * eps-predictor `Model`, with the frozen VAE (diff_discrete.py:70-85) as a bound DnVaeTrain whose decoder only passes data
 * gradients.  Flat packed buffers / stages / refresh exactly as for the VAE engine above; the conditioning path (time MLP and
 * the FiLM / adaptive-norm projections) is stored and differentiated in fp32 in every mode.  The caller supplies what the
 * reference draws or looks up on the host: t, the posterior sample z of the frozen encoder (dn_vae_encode_params +
 * dn_posterior_sample), the beta_0 jitter, the target noise, the fp32 schedule tables and min(snr, 5) / snr per sample.   */
typedef struct DnEpsTrain DnEpsTrain;
typedef struct {
  const float* feat;        /* [B, T, dim_feat] target features (multitask reconstruction loss), may be NULL without multitask */
  const int32_t* units;     /* [B, T] dictionary indices, 0 = pad                                                         */
  const int32_t* lengths;   /* [B]                                                                                        */
  const float* z;           /* [B, T, latent] posterior sample of the frozen VAE encoder (:1530)                          */
  const float* jitter;      /* [B, T, latent]  x1 = z + jitter * beta0 (:1534-1536)                                       */
  const float* true_noise;  /* [B, T, latent]                                                                             */
  const int32_t* times;     /* [B] in [1, timesteps)                                                                      */
  const float* sqrt_ac;     /* fp32 [timesteps] sqrt(alphas_cumprod)                                                      */
  const float* sqrt_1mac;   /* fp32 [timesteps] sqrt(1 - alphas_cumprod)                                                  */
  const float* snr_weight;  /* fp32 [B]: min(snr_t, 5) / snr_t (:1565-1569)                                               */
  float beta0;
  int32_t B, T;
  int32_t n_units;          /* number of non-pad units (the LS-CE divisor, :1594)                                         */
  int32_t n_frames;         /* number of valid frames (the masked recon MSE counts n_frames * dim_feat elements)          */
  int32_t timesteps, multitask;
  float label_smoothing;    /* 0.1 */
  float recon_weight;       /* 50  */
  float loss_scale;
  float* stats;             /* fp32 [8] out: total_loss, nll_loss, recon_mse_loss, noise_loss, acc, n_units, 0, 0          */
  float* eps_out;           /* optional fp32 [B, T, latent]: the predicted noise                                          */
  float attn_dropout;       /* as in DnVaeTrainBatch; applies to the eps-predictor only: the frozen VAE runs in eval mode (:1530) */
  uint32_t dropout_seed_lo, dropout_seed_hi;
  int32_t pad_;
} DnEpsTrainBatch;

int dn_eps_train_create(const DnEpsConfig* cfg, DnEpsTrain** out);
void dn_eps_train_destroy(DnEpsTrain* m);
int64_t dn_eps_train_param_count(const DnEpsTrain* m);
size_t dn_eps_train_aux_bytes(const DnEpsTrain* m);
int dn_eps_train_offsets(const DnEpsTrain* m, int64_t* offsets, int32_t capacity);  /* 21 + 8 * depth entries */
int dn_eps_train_stage_range(const DnEpsTrain* m, int32_t stage, int64_t* offset, int64_t* count);
/* pos_table: fp32 [max_pos + 1, padk(dim)] sinusoidal table (row 0 zeros), a constant of the model                      */
int dn_eps_train_bind(DnEpsTrain* m, float* master, void* work, void* aux, float* grads, const float* pos_table);
int dn_eps_train_refresh(DnEpsTrain* m, void* stream);
size_t dn_eps_train_workspace_bytes(const DnEpsTrain* m, const DnVaeTrain* vae, int32_t B, int32_t T);
int dn_eps_train_forward(DnEpsTrain* m, DnVaeTrain* vae, const DnEpsTrainBatch* batch, void* workspace, size_t workspace_bytes, void* stream);
/* stages: 0 = loss gradients + frozen VAE decoder + final_proj + to_pred, 1 .. depth = transformer layers depth-1 .. 0,
 * depth+1 = WaveNet + init_conv, depth+2 = the conditioning path                                                         */
int dn_eps_train_backward(DnEpsTrain* m, DnVaeTrain* vae, const DnEpsTrainBatch* batch, int32_t first_stage, int32_t last_stage,
                          void* workspace, size_t workspace_bytes, void* stream);

const char* dn_last_error(void);
int dn_version(void);

/* Run-time options (process-wide; no reference counterpart: the reference has no such switches).  Each option starts from the
 * environment variable named beside it, read once when the library first looks, and is changed afterwards only through this entry
 * (the library never re-reads the environment per launch and callers never have to mutate it).  DN_OPTION_DEFAULT restores the
 * start value.  Names: "taps_inner" (DN_TAPS_INNER: K order of a causal conv's taps -- 0 term-outer everywhere, 1 tap-inner on
 * the 256-row tiles [default], 2 tap contractions routed to those tiles by SHAPE whatever the batch size: a batch and its shards
 * then agree bit for bit), "fuse_norm" (DN_FUSE_NORM), "no_split_norm" (DN_NO_SPLIT_NORM), "kblock" (DN_KBLOCK: 0 never / 1 always
 * K-blocked operands), "wgrad_stream", "wgrad_tn", "wgrad_groups" (DN_WGRAD_*: A/B switches of the weight-gradient path).
 * dn_get_option: *is_set = 0 when the option is neither set nor in the environment (the library's built-in choice applies). */
#define DN_OPTION_DEFAULT (-2147483647 - 1)
int dn_set_option(const char* name, int32_t value);
int dn_get_option(const char* name, int32_t* value, int32_t* is_set);

/* Classifier-free guidance (Model.forward_with_cond_scale, latent_module.py:813-826) over ONE pass of twice the batch: `both` fp32
 * [2, n] = the conditioned predictions followed by the null-conditioned ones -> out[i] = null + (cond - null) * scale.           */
int dn_cfg_combine(const float* both, float scale, int64_t n, float* out, void* stream);

/* ------------------------------------------------------------------ the model inside the mask-predict loop */
/* The DECODER side of the reference's NAR S2UT model NARS2UTTransformerModel (research/TranSpeech/nar_transformer.py:569-976; task
 * speech_to_speech_fasttranslate, fairseq/tasks/nat_s2s_task.py:107-127), which the research IterativeRefinementGenerator drives
 * (research/TranSpeech/iterative_refinement_generator.py:131-160).  The speech encoder is out of scope: its output is a given
 * tensor.  Packed tensors (diffnorm_amd/nar_decoder.py::pack_nar; matrices [rows -> 128][K] in cfg.dtype, biases / norms fp32):
 *   emb fp32 [V, D]   pos fp32 [max_pos, D] (sinusoidal, row `pad` zero)   len_W fp32 [256, D] (embed_length.weight)
 *   per layer, stacked: qkv_W [3D] + qkv_b (q ; k ; v of self_attn), so_W / so_b (out_proj), cq_W / cq_b (encoder_attn.q_proj),
 *   ckv_W [2D] / ckv_b (encoder_attn k ; v), co_W / co_b, fc1_W [F] / fc1_b, fc2_W / fc2_b, ln_g / ln_b [3, D] (self_attn_layer_norm,
 *   encoder_attn_layer_norm, final_layer_norm);  fin_g / fin_b (decoder.layer_norm);  out_W [V -> 128-row multiple, D].      */
typedef struct DnNar DnNar;
typedef struct {
  int32_t dim, ffn, layers, heads, vocab, max_pos, pad, dtype;
} DnNarConfig;
int dn_nar_create(const DnNarConfig* cfg, const void* const* weights, int32_t n_weights, DnNar** out);
void dn_nar_destroy(DnNar* m);
size_t dn_nar_workspace_bytes(const DnNar* m, int32_t B, int32_t T, int32_t S);
size_t dn_nar_cross_kv_bytes(const DnNar* m, int32_t B, int32_t S);
/* Keys / values of every layer's encoder attention (fairseq/modules/transformer_layer.py:455-470), which depend on the encoder
 * output only: enc_out fp32 [B, S, D] (batch-major) -> ckv [layers][B*S][k(D) ; v(D)] in cfg.dtype (fp32 for DN_BF16X3), computed
 * once per utterance batch and reused by every refinement iteration.                                                        */
int dn_nar_cross_kv(DnNar* m, const float* enc_out, int32_t B, int32_t S, void* ckv, void* workspace, size_t workspace_bytes, void* stream);
/* forward_length + forward_length_prediction (nar_transformer.py:436-480, no offset): masked mean of the encoder output over the
 * src_lengths[b] valid frames -> embed_length projection (fp32) -> arg-max.  lengths int32 [B].                              */
int dn_nar_predict_lengths(DnNar* m, const float* enc_out, const int32_t* src_lengths, int32_t B, int32_t S, int32_t* lengths,
                           void* workspace, size_t workspace_bytes, void* stream);
/* TransformerUnitDecoder.forward (nar_transformer.py:321-420, inference): tokens int32 [B, T] (`pad` after each row's tokens) ->
 * logits fp32 [B, T, vocab]; dn_cmlm_step turns them into the mask-predict update of forward_decoder (:791-842).            */
int dn_nar_decoder_forward(DnNar* m, const int32_t* tokens, const void* ckv, const int32_t* src_lengths, int32_t B, int32_t T, int32_t S,
                           float* logits, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ the speech encoder of the same model */
/* S2STransformerEncoder (research/TranSpeech/nar_transformer.py:40-76; no speaker embedding) = S2TTransformerEncoder._forward
 * (fairseq/models/speech_to_text/s2t_transformer.py:345-373): Conv1dSubsampler (two Conv1d(kernel, stride 2, padding kernel / 2) + GLU,
 * fairseq/models/speech_to_text/modules/convolution.py:13-57) -> sqrt(dim) scaling + sinusoidal positions of the padding mask -> `layers`
 * pre-norm TransformerEncoderLayers (fairseq/modules/transformer_layer.py:163-226) -> LayerNorm.  Packed tensors
 * (diffnorm_amd/nar_decoder.py::pack_nar_encoder; matrices [rows -> 128][K] in cfg.dtype, biases / norms / positions fp32):
 *   0 conv0_W [conv_channels][padk(kernel * input_dim)] (column j * input_dim + c = weight[o][c][j])   1 conv0_b
 *   2 conv1_W [2 dim][kernel * conv_channels / 2]   3 conv1_b   4 positions [max_pos][dim] (row `pad` zero)
 *   5 qkv_W [layers][3 dim][dim] (q ; k ; v)   6 qkv_b   7 so_W   8 so_b   9 fc1_W   10 fc1_b   11 fc2_W   12 fc2_b
 *   13 ln_g [layers][2][dim] (self_attn_layer_norm ; final_layer_norm)   14 ln_b   15 fin_g   16 fin_b                      */
typedef struct DnNarEnc DnNarEnc;
typedef struct {
  int32_t input_dim, conv_channels, kernel, dim, ffn, layers, heads, max_pos, pad, dtype;
} DnNarEncConfig;
int dn_nar_encoder_create(const DnNarEncConfig* cfg, const void* const* weights, int32_t n_weights, DnNarEnc** out);
void dn_nar_encoder_destroy(DnNarEnc* m);
int32_t dn_nar_encoder_out_frames(int32_t L); /* frames after the two stride-2 layers: floor((L - 1) / 2) + 1, twice */
size_t dn_nar_encoder_workspace_bytes(const DnNarEnc* m, int32_t B, int32_t L);
/* feats fp32 [B, L, input_dim] (zero-padded behind each utterance, as the collater leaves them; the whole padded batch is convolved,
 * like upstream), src_lengths int32 [B] -> enc_out fp32 [B, S, dim] (BATCH-major; upstream returns [S, B, dim]), out_lengths int32 [B]
 * (Conv1dSubsampler.get_out_seq_lens_tensor: the key-padding mask of the layers and of the decoder's encoder attention).          */
int dn_nar_encoder_forward(DnNarEnc* m, const float* feats, const int32_t* src_lengths, int32_t B, int32_t L, float* enc_out,
                           int32_t* out_lengths, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFNORM_HIP_H */
