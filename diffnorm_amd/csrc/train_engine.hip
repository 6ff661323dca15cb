// Training step of the speech VAE on the HIP path (SURVEY 8 f2 / BASELINE config 4): forward with every activation the
// backward needs kept in the caller's workspace, loss + loss gradients, and the backward pass into one flat fp32 gradient
// buffer laid out exactly like the flat parameter buffer (so the optimizer step, the gradient norm and the data-parallel
// all-reduce are single passes over one buffer).
//   reference: SpeechVAEEncoderDecoder.forward latent_module.py:1118-1142, the criterion algebra of
//   fairseq/criterions/speech_vae_decoder_loss.py:60-95, WavenetEncoder :1003-1032, ConditionableTransformer :642-706.
//
// Parameters live in the PACKED layout the contractions consume (engine.h; rows padded to 128, K to 64, the GEGLU projection
// interleaved, conv taps as separate matrices): `master` (fp32) is what the optimizer updates, `work` the same buffer in the
// arithmetic dtype (bf16: written by dn_adam_step's bf16 copy; f32: == master), `aux` holds what dn_vae_train_refresh derives
// after every update (the transposed matrices of the data-gradient contractions, the summed skip biases).  Every contraction,
// forward and backward, is dn_conv_gemm:
//   data gradient    dX[t] = sum_j dY[t + shift_j] W_j            negative shifts, transposed weights
//   weight gradient  dW_j  = dY^T shift_j(X)                      both operands transposed to channels-major K-slices,
//                                                                 grouped split-K contraction, fixed-order reduction
// Everything is enqueued on one stream, allocates nothing and never synchronises.
#include <new>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "common.h"
#include "engine.h"

using namespace dn;

namespace dn {

// dst[(j / chunk)][row0 + c][j % chunk] = src[b*T + t, c] with j = b*Tp + front + t; zero for pad frames, rows >= C and the
// tail columns [B*Tp, cols_total) that round the frame index up to whole K-slices.  64 frames x 64 channels per workgroup
// through LDS with 16-byte global accesses on both sides: loads run along the channels of one frame, stores along the frames
// of one channel (a 64-frame tile never straddles a K-slice: chunk % 64 == 0).  src rows are 16-byte aligned with ld a
// multiple of the access width and >= C rounded up to it (activation buffers: ld = padk(C)).
template <typename T>
__global__ __launch_bounds__(256) void transpose_slices_kernel(const T* __restrict__ src, int ld, int B, int Tn, int C, int front, int Tp,
                                                               int64_t cols_total, T* __restrict__ dst, int rows, int rows_total, int row0,
                                                               int chunk) {
  constexpr int V = 16 / (int)sizeof(T);   // elements per 16-byte access
  constexpr int LDT = 64 + V;              // LDS row: 64 frames + one access of padding (rows stay 16-byte aligned)
  constexpr int CPR = 64 / V;              // 16-byte pieces per 64-element row
  __shared__ __attribute__((aligned(16))) T tile[64][LDT];  // [channel][frame]
  // 8-frame blocks of a channel row are XOR-swizzled by the channel's 8-block, so the scalar writes of one access (lanes 8
  // channels apart) land in different banks; 16-byte reads stay contiguous
  auto swz = [](int c, int f) { return (((f >> 3) ^ ((c >> 3) & 7)) << 3) | (f & 7); };
  const int64_t cols = (int64_t)B * Tp;
  const int64_t j0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < 64 * CPR; i += 256) {  // load: row r = frame, piece q = channels c0 + q*V ..
    const int r = i / CPR, q = i - r * CPR;
    const int64_t j = j0 + r;
    union { uint4 u; T e[V]; } v;
    v.u = make_uint4(0, 0, 0, 0);
    if (j < cols) {
      const int b = (int)(j / Tp), t = (int)(j - (int64_t)b * Tp) - front;
      const int c = c0 + q * V;
      if (t >= 0 && t < Tn && c < C) {
        v.u = *reinterpret_cast<const uint4*>(src + ((int64_t)b * Tn + t) * ld + c);
#pragma unroll
        for (int e = 0; e < V; ++e)
          if (c + e >= C) v.e[e] = T(0);
      }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) tile[q * V + e][swz(q * V + e, r)] = v.e[e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * CPR; i += 256) {  // store: row r = channel, piece q = frames j0 + q*V ..
    const int r = i / CPR, q = i - r * CPR;
    const int c = c0 + r;
    const int64_t j = j0 + q * V;
    if (c < rows && j < cols_total)
      *reinterpret_cast<uint4*>(dst + ((j / chunk) * rows_total + row0 + c) * chunk + j % chunk) = *reinterpret_cast<const uint4*>(&tile[r][swz(r, q * V)]);
  }
}

__global__ void vae_loss_kernel(const float* __restrict__ sums, float* __restrict__ stats, float w_lsce, float w_mse, float w_kl,
                                float eps, int V, float inv_ntokens, float inv_mse, float inv_kl) {
  // sums: [0..3] = sum over frames of {nll, smooth, correct, valid}; [4] = sum of squared errors; [5] = sum of kl rows
  const float eps_i = eps / (float)(V - 1);
  const float lsce = (1.0f - eps - eps_i) * sums[0] + eps_i * sums[1];
  const float nll = sums[0] * inv_ntokens, mse = sums[4] * inv_mse, kl = sums[5] * inv_kl;
  stats[0] = w_lsce * lsce * inv_ntokens + w_mse * mse + w_kl * kl;  // speech_vae_decoder_loss.py:80-83
  stats[1] = nll;
  stats[2] = mse;
  stats[3] = kl;
  stats[4] = sums[3] > 0.f ? sums[2] / sums[3] : 0.f;  // acc
  stats[5] = sums[3];
  stats[6] = lsce * inv_ntokens;
  stats[7] = 0.f;
}

}  // namespace dn

namespace {

inline int64_t r64(int64_t n) { return (n + 63) / 64 * 64; }
inline int64_t take(int64_t& cur, int64_t n) {
  const int64_t o = cur;
  cur += r64(n);
  return o;
}

// ------------------------------------------------------------------------------------------ parameter layout
struct WaveP {
  int cin, cout, S, L;
  int film_col;  // FiLM (eps-predictor): first column of block (stack 0, layer 0)'s [gamma ; beta] in the conditioning rows, -1 = none
  int64_t init_W, init_b, conv_W, conv_b, res_W, res_b, skip_W, skip_b, final_W, final_b;  // elements in master/work/grads
  int64_t t_init, t_conv, t_res, t_skip, t_final;                                          // elements in the transposed region
  int64_t skip_bsum;                                                                       // floats in the derived-fp32 region
};

// The transformer's parameters are stored LAYER-major ([layer][qkv_W, out_W, ffin_W, ffin_b, ffconv_W, ffconv_b, ffout_W,
// ffout_b, g1, g2]) so that the gradients of a layer are one contiguous range that is complete -- and can enter the
// all-reduce -- as soon as that layer's backward has run.  X(l) = element offset of tensor X of layer l.
struct TfP {
  int dim, depth, heads, dim_head, inner;
  int cond_col;  // adaptive norms (eps-predictor): first column of (layer 0, attention norm)'s [gamma ; beta] in the conditioning rows; -1 = learned gammas
  int64_t layer0, layer_stride;
  int64_t o_qkv, o_out, o_ffin, o_ffin_b, o_ffconv, o_ffconv_b, o_ffout, o_ffout_b, o_g1, o_g2;
  int64_t pred_gamma, pred_W;
  int64_t t_qkv, t_out, t_ffin, t_ffconv, t_ffout, t_pred;
  int64_t qkv_W(int l) const { return layer0 + l * layer_stride + o_qkv; }
  int64_t out_W(int l) const { return layer0 + l * layer_stride + o_out; }
  int64_t ffin_W(int l) const { return layer0 + l * layer_stride + o_ffin; }
  int64_t ffin_b(int l) const { return layer0 + l * layer_stride + o_ffin_b; }
  int64_t ffconv_W(int l) const { return layer0 + l * layer_stride + o_ffconv; }
  int64_t ffconv_b(int l) const { return layer0 + l * layer_stride + o_ffconv_b; }
  int64_t ffout_W(int l) const { return layer0 + l * layer_stride + o_ffout; }
  int64_t ffout_b(int l) const { return layer0 + l * layer_stride + o_ffout_b; }
  int64_t g1(int l) const { return layer0 + l * layer_stride + o_g1; }
  int64_t g2(int l) const { return layer0 + l * layer_stride + o_g2; }
};

void layout_wave(WaveP& w, int64_t& cur, int64_t& tcur, int64_t& fcur) {
  const int64_t cinp = padk(w.cin), cp = padk(w.cout), cn = padn(w.cout), S = w.S, L = w.L;
  w.init_W = take(cur, 3 * cn * cinp); w.init_b = take(cur, cp);
  w.conv_W = take(cur, S * L * 3 * cn * cp); w.conv_b = take(cur, S * L * cp);
  w.res_W = take(cur, S * L * cn * cp); w.res_b = take(cur, S * L * cp);
  w.skip_W = take(cur, L * cn * cp); w.skip_b = take(cur, L * cp);
  w.final_W = take(cur, cn * cp); w.final_b = take(cur, cp);
  w.t_init = take(tcur, 3 * (int64_t)padn(cinp) * cp);
  w.t_conv = take(tcur, S * L * 3 * (int64_t)padn(cp) * cp);
  w.t_res = take(tcur, S * L * (int64_t)padn(cp) * cp);
  w.t_skip = take(tcur, L * (int64_t)padn(cp) * cp);
  w.t_final = take(tcur, (int64_t)padn(cp) * cp);
  w.skip_bsum = take(fcur, cp);
}

void layout_tf(TfP& w, int64_t& cur, int64_t& tcur) {
  const int64_t D = w.dim, Dp = padk(D), Dn = padn(D), hd = w.heads * w.dim_head, ip = padk(w.inner), in_n = padn(w.inner), d = w.depth;
  int64_t o = 0;
  w.o_qkv = take(o, padn(3 * hd) * Dp); w.o_out = take(o, Dn * hd);
  w.o_ffin = take(o, 2 * ip * Dp); w.o_ffin_b = take(o, 2 * ip);
  w.o_ffconv = take(o, 3 * in_n * ip); w.o_ffconv_b = take(o, ip);
  w.o_ffout = take(o, Dn * ip); w.o_ffout_b = take(o, Dp);
  w.o_g1 = w.o_g2 = 0;
  if (w.cond_col < 0) {  // conditional norms carry no learned gamma (:662-663)
    w.o_g1 = take(o, D); w.o_g2 = take(o, D);
  }
  w.layer_stride = o;
  w.layer0 = take(cur, d * o);
  w.pred_gamma = take(cur, D);
  w.pred_W = take(cur, Dn * Dp);
  w.t_qkv = take(tcur, d * (int64_t)padn(Dp) * 3 * hd);
  w.t_out = take(tcur, d * (int64_t)padn(hd) * Dp);
  w.t_ffin = take(tcur, d * (int64_t)padn(Dp) * 2 * ip);
  w.t_ffconv = take(tcur, d * 3 * (int64_t)padn(ip) * ip);
  w.t_ffout = take(tcur, d * (int64_t)padn(ip) * Dp);
  w.t_pred = take(tcur, (int64_t)padn(Dp) * Dp);
}

}  // namespace

struct DnVaeTrain {
  DnVaeConfig cfg;
  int n_wave;
  WaveP enc[4], dec[4];
  TfP tf;
  int64_t lm_W, lm_b, t_lm;
  int64_t n_params;   // elements of master / work / grads
  int64_t n_trans;    // elements (arithmetic dtype) of the transposed region of aux
  int64_t n_fderived; // floats of the derived-fp32 region of aux (behind the transposed region)
  float* master;
  void* work;
  char* aux;
  float* grads;
};

namespace {

// Second stream for the weight gradients of a transformer layer (leaves of the backward: nothing downstream reads them before the
// layer's gradient range is handed to the all-reduce).  Their contractions are 144 / 192-tile launches that leave a quarter to
// half of the CUs idle, and each is preceded by its operand transposes; on a stream of their own they run beside the
// data-gradient chain.  Fork: the side stream waits for an event recorded on the main stream when the call is made (its dY operand
// has been enqueued by then).  Join: the main stream waits for the weight gradient's `done` event before it overwrites one of its
// operands, and for the last one at the end of the layer.  One process-wide stream (the shared scratch serialises the weight
// gradients on it); DN_WGRAD_STREAM=0 keeps everything on the caller's stream (read per step).  Measured: VAE update 19.9 -> 18.9 ms,
// diffusion update 36.0 -> 33.9 ms.  The WaveNet stacks' 2 L small weight gradients per stack were tried the same way and LOST
// (19.15 / 35.1 ms with them on the side stream as well: sixteen forks per stack for launches that are too short to matter).
struct WgSide {
  hipStream_t s = nullptr;
  hipEvent_t ready = nullptr, all = nullptr, done[8] = {};
  std::atomic<int> next{0};
  bool tried = false, ok = false;
};
// One side stream and event set PER DEVICE, created (under a lock) on the device that is current when an engine on it first asks:
// an engine on another device of the same process gets its own, never another device's events.
WgSide* wg_side() {
  static WgSide sides[16];
  static std::mutex mu;
  if (option_or(OPT_WGRAD_STREAM, 1) == 0) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  WgSide& side = sides[dev];
  std::lock_guard<std::mutex> lock(mu);
  if (!side.tried) {
    side.tried = true;
    // The weight gradients are filler work beside the data-gradient chain, which is the critical path.  Option wgrad_prio = 1 gives
    // their stream the LOWEST priority the device offers; measured level (VAE update 15.74 vs 15.74 ms, diffusion 27.24 vs 27.19, five
    // alternating pairs): the priority does not change which waiting workgroups the dispatcher places first here.  Default: off.
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);  // (numerically: lowest priority = greatest value)
    const int prio = option_or(OPT_WGRAD_PRIO, 0) != 0 ? prio_lo : 0;
    side.ok = hipStreamCreateWithPriority(&side.s, hipStreamNonBlocking, prio) == hipSuccess &&
              hipEventCreateWithFlags(&side.ready, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&side.all, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 8 && side.ok; ++i) side.ok = hipEventCreateWithFlags(&side.done[i], hipEventDisableTiming) == hipSuccess;
  }
  return side.ok ? &side : nullptr;
}

struct Ctx {
  WgSide* side = nullptr;  // weight gradients that pass a handle run here (transformer layers)
  const float* master;  // fp32 parameters (vectors are read from here in every mode)
  const void* work;     // the same layout in the arithmetic dtype (matrices)
  char* aux;            // transposed matrices, then derived fp32 vectors
  float* grads;
  int64_t n_trans;
  bool frozen;          // no parameter gradients (the frozen VAE decoder inside the diffusion loss): data gradients only
  float attn_dropout = 0.f;  // attention-probability dropout (train mode); layer l hashes with seed_hi + l
  uint32_t seed_lo = 0, seed_hi = 0;
  int dtype, es, B, T, M;
  hipStream_t s;
  void* wg_scratch;     // weight-gradient operands + partial sums
  size_t wg_scratch_bytes = 0;
  float* red_scratch;   // column-sum / norm-backward partials
  const void* W(int64_t off) const { return static_cast<const char*>(work) + off * es; }
  const float* P(int64_t off) const { return master + off; }       // fp32 vectors (biases, gammas)
  float* G(int64_t off) const { return grads + off; }
  const void* Wt(int64_t off) const { return aux + off * es; }
  float* F(int64_t off) const { return reinterpret_cast<float*>(aux + ((n_trans + 63) / 64 * 64) * es) + off; }
};

// ------------------------------------------------------------------------------------------ weight gradient
struct WgPlan {
  int Tp, k_slices, chunk, rows_w, N;
  int64_t cols_total;
  size_t b_dyT, b_xT, b_part;
  size_t total() const { return b_dyT + b_xT + b_part + 768; }
};

WgPlan plan_wgrad(int cin, int cout, int n_taps, int max_shift, int B, int T, int es) {
  WgPlan p;
  p.Tp = round_up(T + max_shift, 64);
  p.rows_w = padn(cin);
  p.N = n_taps * p.rows_w;
  const int64_t cols = (int64_t)B * p.Tp;
  const int64_t tiles = (int64_t)((cout + 255) / 256) * ((p.N + 255) / 256);
  int ks = 1;
  // K-slices of >= 4 K-tiles until about half the chip has a tile-slice; a contraction with >= 160 output tiles runs unsliced and
  // accumulates straight into the gradient (no partial sums to write, re-read and reduce).  Half the chip, not all of it: the
  // sliced contractions move 4.6-5 TB/s through the fabric, most of it their fp32 partial sums (slices x the weight's size, written
  // and read back by the reduction), so halving the slices beats filling the last CUs -- measured on the training bench: target
  // 256 tile-slices 1153 samples/s, 160: 1199, 128: 1207, 64: 1207, 32: 1188, 1 (never slice): 1119.  DN_WGRAD_FILL overrides.
  static const int fill_target = getenv("DN_WGRAD_FILL") ? atoi(getenv("DN_WGRAD_FILL")) : 128;
  static const int unsliced_min = getenv("DN_WGRAD_UNSLICED") ? atoi(getenv("DN_WGRAD_UNSLICED")) : 160;
  while (ks < 64 && tiles * ks < (ks == 1 ? unsliced_min : fill_target) && cols / (ks * 2) >= 256) ks *= 2;
  p.k_slices = ks;
  p.cols_total = (cols + (int64_t)64 * ks - 1) / ((int64_t)64 * ks) * ((int64_t)64 * ks);
  p.chunk = (int)(p.cols_total / ks);
  p.b_dyT = ((size_t)ks * cout * p.chunk * es + 255) & ~size_t(255);
  p.b_xT = ((size_t)ks * p.N * p.chunk * es + 255) & ~size_t(255);
  p.b_part = (size_t)ks * cout * p.N * 4;
  return p;
}

struct WgTap { const void* x; int ldx; int shift; };

// the data gradient of the conditioning projection from the row-major master matrix (dn_rows_times_weight); option
// cond_stream = 0: from the transposed copy with a split-K contraction.  The option is read when the working copies are refreshed
// too: switch it between updates only after a refresh.
inline bool cond_streamed() { return option_or(OPT_COND_STREAM, 1) != 0; }


template <typename T>
void launch_transpose_slices(const void* src, int ld, int B, int Tn, int C, int front, const WgPlan& pl, void* dst, int rows, int rows_total,
                             int row0, hipStream_t s) {
  dim3 grid((unsigned)((pl.cols_total + 63) / 64), (unsigned)((rows + 63) / 64));
  hipLaunchKernelGGL((dn::transpose_slices_kernel<T>), grid, dim3(256), 0, s, static_cast<const T*>(src), ld, B, Tn, C, front, pl.Tp,
                     pl.cols_total, static_cast<T*>(dst), rows, rows_total, row0, pl.chunk);
}

// main stream: wait for weight gradient `handle` (weight_grad's `overlap` result; < 0: it ran on the main stream)
int wg_wait(const Ctx& c, int handle) {
  if (handle < 0 || !c.side) return DN_OK;
  return hipStreamWaitEvent(c.s, c.side->done[handle], 0) == hipSuccess ? DN_OK : DN_ELAUNCH;
}

// grad[tap][Np][Kp] += dY^T . shift_tap(X_tap); dY [M, lddy] (cout valid columns), X_tap [M, ldx] (cin valid columns).
// overlap != nullptr: run on the side stream if there is one and return the handle to wait on (wg_wait) before dY or X is
// overwritten and before the gradient is used; -1 = it ran on c.s.
int weight_grad(const Ctx& c0, const WgTap* taps, int n_taps, int cin, const void* dy, int lddy, int cout, float* grad, int tag = 0,
                int* overlap = nullptr, const WgTnGroups* grp = nullptr) {
  if (overlap) *overlap = -1;
  if (c0.frozen) return DN_OK;
  Ctx c = c0;
  WgSide* const side = overlap ? c0.side : nullptr;
  if (side) {
    if (hipEventRecord(side->ready, c0.s) != hipSuccess || hipStreamWaitEvent(side->s, side->ready, 0) != hipSuccess) return DN_ELAUNCH;
    c.s = side->s;
  }
  auto finish = [&](int rc) {
    if (rc != DN_OK || !side) return rc;
    const int h = side->next.fetch_add(1) & 7;
    if (hipEventRecord(side->done[h], side->s) != hipSuccess) return (int)DN_ELAUNCH;
    *overlap = h;
    return rc;
  };
  int max_shift = 0;
  for (int j = 0; j < n_taps; ++j) max_shift = taps[j].shift > max_shift ? taps[j].shift : max_shift;
  const WgPlan pl = plan_wgrad(cin, cout, n_taps, max_shift, c.B, c.T, c.es);
  char* base = static_cast<char*>(c.wg_scratch);
  // bf16: straight from the row-major operands (wgrad_tn.hip: transposing LDS reads, no channel-major copies).  Slices of the
  // frames until about 160 tile-slices (at least 256 frames each), partial sums in the scratch the transposed copies would use;
  // a contraction that has its tiles anyway accumulates straight into the gradient.  DN_WGRAD_TN=0: the transposed-copies form.
  const bool use_tn = c.es == 2 && option_or(OPT_WGRAD_TN, 1) != 0;
  const bool grouped = option_or(OPT_WGRAD_GROUPS, 1) != 0;  // 0: one launch per block also in the row-major form (A/B timing)
  if (grp && grp->groups > 1 && (!use_tn || !grouped)) {  // the transposed-copies form has no groups: one call per group
    for (int g = 0; g < grp->groups; ++g) {
      WgTap tg[DN_MAX_TERMS];
      for (int j = 0; j < n_taps; ++j)
        tg[j] = WgTap{eoff(taps[j].x, (size_t)g * grp->x_gstride, c.es), taps[j].ldx, grp->shift_by_group ? taps[j].shift << g : taps[j].shift};
      DN_TRY(weight_grad(c0, tg, n_taps, cin, eoff(dy, (size_t)g * grp->dy_gstride, c.es), lddy, cout, grad + g * grp->grad_gstride, tag, nullptr, nullptr));
    }
    return DN_OK;
  }
  if (use_tn) {
    const int rows_w = padn(cin), n_total = n_taps * rows_w, groups = grp ? grp->groups : 1;
    const long tiles = (long)((cout + 255) / 256) * ((n_total + 255) / 256) * groups;
    const size_t cap = c.wg_scratch_bytes > pl.total() ? c.wg_scratch_bytes : pl.total();
    int slices = 1;
    while (slices < 16 && tiles * slices < 160 && (long)c.M / (slices * 2) >= 256 &&
           (size_t)(slices * 2) * groups * cout * n_total * 4 <= cap) slices *= 2;
    if (groups > 1 && slices == 1 && (size_t)groups * cout * n_total * 4 > cap) return DN_EWORKSPACE;  // (cannot happen: max_wgrad_bytes plans it)
    const void* xs[DN_MAX_TERMS]; int ldx[DN_MAX_TERMS], sh[DN_MAX_TERMS];
    for (int j = 0; j < n_taps; ++j) { xs[j] = taps[j].x; ldx[j] = taps[j].ldx; sh[j] = taps[j].shift; }
    float* part = slices > 1 ? reinterpret_cast<float*>(base) : nullptr;
    int rc = wgrad_tn_launch(dy, lddy, cout, xs, ldx, sh, n_taps, cin, c.B, c.T, slices, part, slices > 1 ? nullptr : grad, c.s, tag, grp);
    for (int g = 0; g < groups && rc == DN_OK && slices > 1; ++g)
      rc = dn_wgrad_reduce(part + (size_t)g * slices * cout * n_total, slices, cout, n_total, rows_w, n_taps,
                           grad + (grp ? g * grp->grad_gstride : 0), padn(cout), padk(cin), c.s);
    return finish(rc);
  }
  void* dyT = base;
  void* xT = base + pl.b_dyT;
  float* part = reinterpret_cast<float*>(base + pl.b_dyT + pl.b_xT);
  if (c.es == 2) {
    launch_transpose_slices<uint16_t>(dy, lddy, c.B, c.T, cout, 0, pl, dyT, cout, cout, 0, c.s);
    for (int j = 0; j < n_taps; ++j)
      launch_transpose_slices<uint16_t>(taps[j].x, taps[j].ldx, c.B, c.T, cin, taps[j].shift, pl, xT, pl.rows_w, pl.N, j * pl.rows_w, c.s);
  } else {
    launch_transpose_slices<float>(dy, lddy, c.B, c.T, cout, 0, pl, dyT, cout, cout, 0, c.s);
    for (int j = 0; j < n_taps; ++j)
      launch_transpose_slices<float>(taps[j].x, taps[j].ldx, c.B, c.T, cin, taps[j].shift, pl, xT, pl.rows_w, pl.N, j * pl.rows_w, c.s);
  }
  DN_CHECK_LAUNCH("weight_grad transposes");
  if (pl.k_slices == 1) {  // one group per tap, each accumulating into its own matrix of the packed gradient
    const int Np = padn(cout), Kp = padk(cin);
    DnGemmParams p = gemm_base(c.dtype, cout, Kp, pl.chunk, cout);
    p.groups = n_taps;
    p.terms[0].A = dyT; p.terms[0].lda = pl.chunk; p.terms[0].a_gstride = 0;
    p.terms[0].W = xT; p.terms[0].w_gstride = (int64_t)pl.rows_w * pl.chunk;
    p.epilogue = DN_EPI_RESADD; p.res = grad; p.ldr = Kp; p.res_gstride = (int64_t)Np * Kp;
    p.out = grad; p.ldo = Kp; p.out_dtype = DN_F32; p.out_gstride = (int64_t)Np * Kp;
    p.pad_ = tag << 8;
    return finish(dn_conv_gemm(&p, c.s));
  }
  DnGemmParams p = gemm_base(c.dtype, cout, pl.N, pl.chunk, cout);
  p.groups = pl.k_slices;
  p.terms[0].A = dyT; p.terms[0].lda = pl.chunk; p.terms[0].a_gstride = (int64_t)cout * pl.chunk;
  p.terms[0].W = xT; p.terms[0].w_gstride = (int64_t)pl.N * pl.chunk;
  p.out = part; p.ldo = pl.N; p.out_dtype = DN_F32; p.out_gstride = (int64_t)cout * pl.N;
  p.pad_ = tag << 8;
  DN_TRY(dn_conv_gemm(&p, c.s));
  return finish(dn_wgrad_reduce(part, pl.k_slices, cout, pl.N, pl.rows_w, n_taps, grad, padn(cout), padk(cin), c.s));
}

int bias_grad(const Ctx& c, const void* dy, int ld, int dtype, int groups, int rows_per_group, int C, float* grad, int out_ld) {
  if (c.frozen) return DN_OK;
  return dn_colsum(dy, ld, dtype, groups, rows_per_group, C, grad, out_ld, 1.0f, 1, c.red_scratch, c.s);
}

// ------------------------------------------------------------------------------------------ WaveNet
struct WaveSave {  // activations kept by the forward
  void *h0, *res, *hpre, *out, *sk;  // res / hpre / out: [S][L][M][cp]
};
struct WaveTmp {  // backward temporaries, shared by all WaveNets of a model (sized for the widest)
  void *d_sk, *d_out0, *d_out1, *d_h, *d_h0;
};

WaveSave plan_wave_save(const WaveP& w, int M, int es, Arena& ar) {
  const size_t one = (size_t)M * padk(w.cout) * es, all = one * w.S * w.L;
  WaveSave b;
  b.h0 = ar.take(one); b.res = ar.take(all); b.hpre = ar.take(all); b.out = ar.take(all); b.sk = ar.take(one);
  return b;
}

// `fin` carries the destination of the final 1x1 conv (out, ldo, out_dtype, N).  gb (FiLM only): fp32 conditioning rows [B, gb_ld],
// block (st, i) at columns film_col + (st * L + i) * 2 * cp: [gamma (cp) ; beta (cp)].
int wave_forward(const Ctx& c, const WaveP& w, const void* in, const WaveSave& sv, DnGemmParams fin, const float* gb = nullptr, int gb_ld = 0) {
  const int dtype = c.dtype, es = c.es, M = c.M, T = c.T;
  const int cinp = padk(w.cin), cp = padk(w.cout), cn = padn(w.cout), L = w.L, S = w.S;
  const size_t mat = (size_t)cn * cp;
  const int64_t plane = (int64_t)M * cp;
  {  // init conv k=3 (:1014,1029)
    DnGemmParams p = gemm_base(dtype, M, cp, cinp, T);
    p.n_terms = 3;
    for (int j = 0; j < 3; ++j) {
      p.terms[j].A = in; p.terms[j].lda = cinp; p.terms[j].shift = 2 - j;
      p.terms[j].W = eoff(c.W(w.init_W), (size_t)j * cn * cinp, es);
    }
    p.bias = c.P(w.init_b); p.out = sv.h0; p.ldo = cp;
    DN_TRY(dn_conv_gemm(&p, c.s));
  }
  for (int st = 0; st < S; ++st) {
    const void* in_s = st == 0 ? sv.h0 : eoff(sv.out, (size_t)(st - 1) * L * plane, es);
    const int64_t a_gs = st == 0 ? 0 : plane;
    void* res_s = eoff(sv.res, (size_t)st * L * plane, es);
    void* h_s = eoff(sv.hpre, (size_t)st * L * plane, es);
    void* out_s = eoff(sv.out, (size_t)st * L * plane, es);
    {  // res_conv 1x1 (:510,521)
      DnGemmParams p = gemm_base(dtype, M, cp, cp, T);
      p.groups = L;
      p.terms[0].A = in_s; p.terms[0].lda = cp; p.terms[0].a_gstride = a_gs;
      p.terms[0].W = eoff(c.W(w.res_W), (size_t)st * L * mat, es); p.terms[0].w_gstride = (int64_t)mat;
      p.bias = c.P(w.res_b) + (size_t)st * L * cp; p.bias_gstride = cp;
      p.out = res_s; p.ldo = cp; p.out_gstride = plane;
      DN_TRY(dn_conv_gemm(&p, c.s));
    }
    {  // dilated conv k=3, pre-activation kept (:509,523)
      DnGemmParams p = gemm_base(dtype, M, cp, cp, T);
      p.groups = L;
      p.n_terms = 3;
      for (int j = 0; j < 3; ++j) {
        p.terms[j].A = in_s; p.terms[j].lda = cp; p.terms[j].a_gstride = a_gs;
        p.terms[j].shift = 2 - j; p.terms[j].shift_by_group = 1;
        p.terms[j].W = eoff(c.W(w.conv_W), ((size_t)st * L * 3 + j) * mat, es); p.terms[j].w_gstride = (int64_t)(3 * mat);
      }
      p.bias = c.P(w.conv_b) + (size_t)st * L * cp; p.bias_gstride = cp;
      p.out = h_s; p.ldo = cp; p.out_gstride = plane;
      DN_TRY(dn_conv_gemm(&p, c.s));
    }
    // [FiLM h * gamma + beta (:517-527);] tanh(h) * sigmoid(h) + res (:528-530): all L blocks in one pass, or one per block
    // when each has its own conditioning columns
    if (!gb || w.film_col < 0) {
      DN_TRY(dn_gate_forward(h_s, res_s, out_s, dtype, L * M, cp, T, nullptr, 0, 0, c.s));
    } else {
      for (int i = 0; i < L; ++i)
        DN_TRY(dn_gate_forward(eoff(h_s, (size_t)i * plane, es), eoff(res_s, (size_t)i * plane, es), eoff(out_s, (size_t)i * plane, es), dtype, M,
                               cp, T, gb + w.film_col + (size_t)(st * L + i) * 2 * cp, gb_ld, cp, c.s));
    }
  }
  const void* last = eoff(sv.out, (size_t)(S - 1) * L * plane, es);
  {  // sum over blocks of skip_conv(out_i) (:511,534,617)
    DnGemmParams p = gemm_base(dtype, M, cp, cp, T);
    p.n_terms = L;
    for (int i = 0; i < L; ++i) {
      p.terms[i].A = eoff(last, (size_t)i * plane, es); p.terms[i].lda = cp;
      p.terms[i].W = eoff(c.W(w.skip_W), (size_t)i * mat, es);
    }
    p.bias = c.F(w.skip_bsum); p.out = sv.sk; p.ldo = cp;
    DN_TRY(dn_conv_gemm(&p, c.s));
  }
  fin.dtype = dtype; fin.M = M; fin.K = cp; fin.T = T; fin.groups = 1; fin.n_terms = 1;
  fin.terms[0].A = sv.sk; fin.terms[0].lda = cp; fin.terms[0].W = c.W(w.final_W);
  fin.bias = c.P(w.final_b);
  return dn_conv_gemm(&fin, c.s);
}

// d_y: gradient w.r.t. the final conv's output, arithmetic dtype [M, padk(cout)] (pad columns zero).
// d_x (optional): gradient w.r.t. the WaveNet's input, [M, padk(cin)] in d_x_dtype.
// FiLM: gb as in wave_forward; d_gb (fp32 [B, gb_ld], same columns) accumulates the conditioning gradients, film_rows is a
// [M, 2 * cp] fp32 temporary for the per-frame terms.
int wave_backward(const Ctx& c, const WaveP& w, const void* in, const WaveSave& sv, const void* d_y, void* d_x, int d_x_dtype,
                  const WaveTmp& tb, const float* gb = nullptr, int gb_ld = 0, float* d_gb = nullptr, float* film_rows = nullptr) {
  const int dtype = c.dtype, es = c.es, M = c.M, T = c.T;
  const int cinp = padk(w.cin), cp = padk(w.cout), cn = padn(w.cout), L = w.L, S = w.S;
  const size_t mat = (size_t)cn * cp, tmat = (size_t)padn(cp) * cp;
  const int64_t plane = (int64_t)M * cp;
  // final conv
  {
    WgTap tap{sv.sk, cp, 0};
    DN_TRY(weight_grad(c, &tap, 1, w.cout, d_y, cp, w.cout, c.G(w.final_W)));
    DN_TRY(bias_grad(c, d_y, cp, dtype, 1, M, cp, c.G(w.final_b), 0));
    DnGemmParams p = gemm_base(dtype, M, cp, cp, T);
    p.terms[0].A = d_y; p.terms[0].lda = cp; p.terms[0].W = c.Wt(w.t_final);
    p.out = tb.d_sk; p.ldo = cp;
    DN_TRY(dn_conv_gemm(&p, c.s));
  }
  const void* last = eoff(sv.out, (size_t)(S - 1) * L * plane, es);
  {  // skip convs of the last stack: one weight-gradient contraction for all L blocks (dY shared), one bias gradient
    WgTap taps[DN_MAX_TERMS];
    for (int i = 0; i < L; ++i) taps[i] = WgTap{eoff(last, (size_t)i * plane, es), cp, 0};
    DN_TRY(weight_grad(c, taps, L, w.cout, tb.d_sk, cp, w.cout, c.G(w.skip_W)));
    // every skip bias sees the same gradient: column sums once, then added to each of the L rows
    if (!c.frozen) {
      float* tmp = c.red_scratch + 1024 * 1024;  // behind the partial sums of dn_colsum
      DN_TRY(dn_colsum(tb.d_sk, cp, dtype, 1, M, cp, tmp, 0, 1.0f, 0, c.red_scratch, c.s));
      DN_TRY(dn_add_broadcast(tmp, c.G(w.skip_b), cp, cp, L, c.s));
    }
    DnGemmParams p = gemm_base(dtype, M, cp, cp, T);  // d out_i = d sk . W_skip_i
    p.groups = L;
    p.terms[0].A = tb.d_sk; p.terms[0].lda = cp; p.terms[0].a_gstride = 0;
    p.terms[0].W = c.Wt(w.t_skip); p.terms[0].w_gstride = (int64_t)tmat;
    p.out = tb.d_out0; p.ldo = cp; p.out_gstride = plane;
    DN_TRY(dn_conv_gemm(&p, c.s));
  }
  void* d_out = tb.d_out0;
  void* d_next = tb.d_out1;
  for (int st = S - 1; st >= 0; --st) {
    const void* in_s = st == 0 ? sv.h0 : eoff(sv.out, (size_t)(st - 1) * L * plane, es);
    const void* h_s = eoff(sv.hpre, (size_t)st * L * plane, es);
    // d h = d out * gate'(h) (:528) [* gamma, and the conditioning gradients: sum over t of dg * h and dg per sample]
    if (!gb || w.film_col < 0) {
      DN_TRY(dn_gate_backward(d_out, h_s, tb.d_h, dtype, L * M, cp, T, nullptr, 0, 0, nullptr, 0, c.s));
    } else {
      for (int i = 0; i < L; ++i) {
        const size_t col = w.film_col + (size_t)(st * L + i) * 2 * cp;
        DN_TRY(dn_gate_backward(eoff(d_out, (size_t)i * plane, es), eoff(h_s, (size_t)i * plane, es), eoff(tb.d_h, (size_t)i * plane, es), dtype,
                                M, cp, T, gb + col, gb_ld, cp, film_rows, 2 * cp, c.s));
        DN_TRY(dn_colsum(film_rows, 2 * cp, DN_F32, c.B, T, 2 * cp, d_gb + col, gb_ld, 1.0f, 1, c.red_scratch, c.s));
      }
    }
    DN_TRY(bias_grad(c, d_out, cp, dtype, L, M, cp, c.G(w.res_b) + (size_t)st * L * cp, cp));
    DN_TRY(bias_grad(c, tb.d_h, cp, dtype, L, M, cp, c.G(w.conv_b) + (size_t)st * L * cp, cp));
    {  // the L blocks of the stack as groups of one launch each (block i: input plane i -- stack 0: the shared input --, dilation 2^i)
      WgTap r{in_s, cp, 0};
      const WgTnGroups gr{L, plane, st == 0 ? 0 : plane, (int64_t)mat, 0};
      DN_TRY(weight_grad(c, &r, 1, w.cout, d_out, cp, w.cout, c.G(w.res_W) + (size_t)st * L * mat, 0, nullptr, &gr));
      WgTap taps[3];
      for (int j = 0; j < 3; ++j) taps[j] = WgTap{in_s, cp, 2 - j};
      const WgTnGroups gc{L, plane, st == 0 ? 0 : plane, (int64_t)(3 * mat), 1};
      DN_TRY(weight_grad(c, taps, 3, w.cout, tb.d_h, cp, w.cout, c.G(w.conv_W) + (size_t)st * L * 3 * mat, 0, nullptr, &gc));
    }
    {  // d in_i = d out_i . W_res_i + sum_j d h_i[t + (2-j) 2^i] . W_conv_i,j   (one grouped contraction, 4 terms)
      DnGemmParams p = gemm_base(dtype, M, cp, cp, T);
      p.groups = L;
      p.n_terms = 4;
      p.terms[0].A = d_out; p.terms[0].lda = cp; p.terms[0].a_gstride = plane;
      p.terms[0].W = eoff(c.Wt(w.t_res), (size_t)st * L * tmat, es); p.terms[0].w_gstride = (int64_t)tmat;
      for (int j = 0; j < 3; ++j) {
        DnGemmTerm& t = p.terms[1 + j];
        t.A = tb.d_h; t.lda = cp; t.a_gstride = plane;
        t.shift = -(2 - j); t.shift_by_group = 1;
        t.W = eoff(c.Wt(w.t_conv), ((size_t)st * L * 3 + j) * tmat, es); t.w_gstride = (int64_t)(3 * tmat);
      }
      p.out = d_next; p.ldo = cp; p.out_gstride = plane;
      DN_TRY(dn_conv_gemm(&p, c.s));
    }
    void* t = d_out; d_out = d_next; d_next = t;
  }
  // stack 0 fed one tensor to all blocks (:570-571): its gradient is the sum over blocks
  DN_TRY(dn_sum_groups(d_out, plane, L, tb.d_h0, dtype, plane, c.s));
  {
    WgTap taps[3];
    for (int j = 0; j < 3; ++j) taps[j] = WgTap{in, cinp, 2 - j};
    DN_TRY(weight_grad(c, taps, 3, w.cin, tb.d_h0, cp, w.cout, c.G(w.init_W)));
    DN_TRY(bias_grad(c, tb.d_h0, cp, dtype, 1, M, cp, c.G(w.init_b), 0));
  }
  if (!d_x) return DN_OK;
  DnGemmParams p = gemm_base(dtype, M, cinp, cp, T);
  p.n_terms = 3;
  for (int j = 0; j < 3; ++j) {
    p.terms[j].A = tb.d_h0; p.terms[j].lda = cp; p.terms[j].shift = -(2 - j);
    p.terms[j].W = eoff(c.Wt(w.t_init), (size_t)j * padn(cinp) * cp, es);
  }
  p.out = d_x; p.ldo = cinp; p.out_dtype = d_x_dtype;
  return dn_conv_gemm(&p, c.s);
}

// ------------------------------------------------------------------------------------------ transformer (learned-gamma norms)
struct TfSave {  // per layer l: x [depth+1][M][Dp] fp32, xmid [depth][M][Dp] fp32, and the operand copies
  float *x, *xmid, *lse;
  void *xn1, *qkv, *ao, *xn2, *pre, *gg, *fc, *xnp;
};
struct TfTmp {
  float* dx;       // [M][Dp] fp32 residual-stream gradient (updated in place)
  void *dx_act, *d_fc, *d_gg, *d_pre, *d_xn, *d_ao, *d_qkv;
  float* delta;
};

TfSave plan_tf_save(const TfP& w, int B, int T, int es, Arena& ar) {
  const size_t M = (size_t)B * T, Dp = padk(w.dim), hd = w.heads * w.dim_head, ip = padk(w.inner), d = w.depth;
  TfSave s;
  s.x = (float*)ar.take((d + 1) * M * Dp * 4);
  s.xmid = (float*)ar.take(d * M * Dp * 4);
  s.lse = (float*)ar.take(d * (size_t)B * w.heads * T * 4);
  s.xn1 = ar.take(d * M * Dp * es); s.qkv = ar.take(d * M * 3 * hd * es); s.ao = ar.take(d * M * hd * es);
  s.xn2 = ar.take(d * M * Dp * es); s.pre = ar.take(d * M * 2 * ip * es); s.gg = ar.take(d * M * ip * es);
  s.fc = ar.take(d * M * ip * es); s.xnp = ar.take(M * Dp * es);
  return s;
}

TfTmp plan_tf_tmp(const TfP& w, int B, int T, int es, Arena& ar) {
  const size_t M = (size_t)B * T, Dp = padk(w.dim), hd = w.heads * w.dim_head, ip = padk(w.inner);
  TfTmp t;
  t.dx = (float*)ar.take(M * Dp * 4);
  t.dx_act = ar.take(M * Dp * es); t.d_fc = ar.take(M * ip * es); t.d_gg = ar.take(M * ip * es); t.d_pre = ar.take(M * 2 * ip * es);
  t.d_xn = ar.take(M * Dp * es); t.d_ao = ar.take(M * hd * es); t.d_qkv = ar.take(M * 3 * hd * es);
  t.delta = (float*)ar.take((size_t)B * w.heads * T * 4);
  return t;
}

// x[0] holds the input residual stream; `pred` receives to_pred's output (fp32 [M, pred_ld]).
// gb (adaptive norms only): conditioning rows [B, gb_ld]; norm (l, j) at columns cond_col + (2 l + j) * 2 * Dp.
int tf_forward(const Ctx& c, const TfP& w, const int32_t* lengths, const TfSave& sv, float* pred, int pred_ld, const float* gb = nullptr,
               int gb_ld = 0) {
  const int dtype = c.dtype, es = c.es, B = c.B, T = c.T, M = c.M;
  const int D = w.dim, Dp = padk(D), Dn = padn(D), hd = w.heads * w.dim_head, ip = padk(w.inner), in_n = padn(w.inner);
  const size_t MD = (size_t)M * Dp;
  for (int l = 0; l < w.depth; ++l) {
    float* x = sv.x + l * MD;
    float* xmid = sv.xmid + l * MD;
    float* xnext = sv.x + (l + 1) * MD;
    void* xn1 = eoff(sv.xn1, l * MD, es);
    void* qkv = eoff(sv.qkv, (size_t)l * M * 3 * hd, es);
    void* ao = eoff(sv.ao, (size_t)l * M * hd, es);
    void* xn2 = eoff(sv.xn2, l * MD, es);
    void* pre = eoff(sv.pre, (size_t)l * M * 2 * ip, es);
    void* gg = eoff(sv.gg, (size_t)l * M * ip, es);
    void* fc = eoff(sv.fc, (size_t)l * M * ip, es);
    const bool ada = gb && w.cond_col >= 0;
    const float* gb1 = ada ? gb + w.cond_col + (size_t)(2 * l) * 2 * Dp : nullptr;
    const float* gb2 = ada ? gb + w.cond_col + (size_t)(2 * l + 1) * 2 * Dp : nullptr;
    DN_TRY(dn_rmsnorm(x, Dp, xn1, Dp, dtype, M, D, T, ada ? nullptr : c.P(w.g1(l)), gb1, gb_ld, Dp, c.s));
    {  // to_q ; to_kv (:930-931,945)
      DnGemmParams p = gemm_base(dtype, M, 3 * hd, Dp, T);
      p.terms[0].A = xn1; p.terms[0].lda = Dp; p.terms[0].W = c.W(w.qkv_W(l));
      p.out = qkv; p.ldo = 3 * hd;
      DN_TRY(dn_conv_gemm(&p, c.s));
    }
    {
      DnAttnParams a;
      memset(&a, 0, sizeof(a));
      a.q = qkv; a.k = eoff(qkv, hd, es); a.v = eoff(qkv, 2 * hd, es); a.out = ao;
      a.ldq = a.ldk = a.ldv = 3 * hd; a.ldo = hd;
      a.B = B; a.T = T; a.heads = w.heads; a.dim_head = w.dim_head; a.dtype = dtype; a.lengths = lengths;
      a.scale = 1.0f / sqrtf((float)w.dim_head);
      a.lse = sv.lse + (size_t)l * B * w.heads * T;
      a.dropout_p = c.attn_dropout; a.seed_lo = c.seed_lo; a.seed_hi = c.seed_hi + (uint32_t)l;
      DN_TRY(dn_attention(&a, c.s));
    }
    {  // to_out + residual (:932,692)
      DnGemmParams p = gemm_base(dtype, M, Dp, hd, T);
      p.terms[0].A = ao; p.terms[0].lda = hd; p.terms[0].W = c.W(w.out_W(l));
      p.epilogue = DN_EPI_RESADD; p.res = x; p.ldr = Dp; p.out = xmid; p.ldo = Dp; p.out_dtype = DN_F32;
      DN_TRY(dn_conv_gemm(&p, c.s));
    }
    DN_TRY(dn_rmsnorm(xmid, Dp, xn2, Dp, dtype, M, D, T, ada ? nullptr : c.P(w.g2(l)), gb2, gb_ld, Dp, c.s));
    if (option_or(OPT_FUSED_GEGLU, 1) != 0) {  // Linear(D -> 2*inner) + GEGLU in the contraction's epilogue as inference runs it (:899,881-884),
      DnGemmParams p = gemm_base(dtype, M, ip, Dp, T);  // which also keeps the pre-activation (packed [8 value ; 8 gate] columns) for the backward
      p.terms[0].A = xn2; p.terms[0].lda = Dp; p.terms[0].W = c.W(w.ffin_W(l));
      p.bias = c.P(w.ffin_b(l)); p.epilogue = DN_EPI_GEGLU; p.out = gg; p.ldo = ip;
      p.pre_out = pre; p.pre_ld = 2 * ip;
      DN_TRY(dn_conv_gemm(&p, c.s));
    } else {  // option fused_geglu = 0: the projection, then a pass over its output (A/B timing)
      DnGemmParams p = gemm_base(dtype, M, 2 * ip, Dp, T);
      p.terms[0].A = xn2; p.terms[0].lda = Dp; p.terms[0].W = c.W(w.ffin_W(l));
      p.bias = c.P(w.ffin_b(l)); p.out = pre; p.ldo = 2 * ip;
      DN_TRY(dn_conv_gemm(&p, c.s));
      DN_TRY(dn_geglu_forward(pre, gg, dtype, M, ip, c.s));
    }
    {  // CausalConv1d(inner, inner, 3) (:894)
      DnGemmParams p = gemm_base(dtype, M, ip, ip, T);
      p.n_terms = 3;
      for (int j = 0; j < 3; ++j) {
        p.terms[j].A = gg; p.terms[j].lda = ip; p.terms[j].shift = 2 - j;
        p.terms[j].W = eoff(c.W(w.ffconv_W(l)), (size_t)j * in_n * ip, es);
      }
      p.bias = c.P(w.ffconv_b(l)); p.out = fc; p.ldo = ip;
      DN_TRY(dn_conv_gemm(&p, c.s));
    }
    {  // Linear(inner -> D) + residual (:902,704)
      DnGemmParams p = gemm_base(dtype, M, Dp, ip, T);
      p.terms[0].A = fc; p.terms[0].lda = ip; p.terms[0].W = c.W(w.ffout_W(l));
      p.bias = c.P(w.ffout_b(l));
      p.epilogue = DN_EPI_RESADD; p.res = xmid; p.ldr = Dp; p.out = xnext; p.ldo = Dp; p.out_dtype = DN_F32;
      DN_TRY(dn_conv_gemm(&p, c.s));
    }
  }
  // to_pred = RMSNorm(gamma) + Linear(D, D, no bias) (:676-679)
  DN_TRY(dn_rmsnorm(sv.x + w.depth * MD, Dp, sv.xnp, Dp, dtype, M, D, T, c.P(w.pred_gamma), nullptr, 0, 0, c.s));
  DnGemmParams p = gemm_base(dtype, M, Dp, Dp, T);
  p.terms[0].A = sv.xnp; p.terms[0].lda = Dp; p.terms[0].W = c.W(w.pred_W);
  p.out = pred; p.ldo = pred_ld; p.out_dtype = DN_F32;
  return dn_conv_gemm(&p, c.s);
}

// data gradient of a Linear: out[M, Kp] = dY[M, K = padk(N)] . W  (Wt = the transposed packed weight [padn(Kp)][padk(N)])
int linear_dgrad(const Ctx& c, const void* dy, int lddy, int Kc, const void* wt, void* out, int n_out, int out_dtype) {
  DnGemmParams p = gemm_base(c.dtype, c.M, n_out, Kc, c.T);
  p.terms[0].A = dy; p.terms[0].lda = lddy; p.terms[0].W = wt;
  p.out = out; p.ldo = n_out; p.out_dtype = out_dtype;
  return dn_conv_gemm(&p, c.s);
}

// to_pred backward: d_pred (arithmetic dtype [M, Dp]) -> tb.dx / tb.dx_act = gradient w.r.t. x[depth]
int tf_backward_head(const Ctx& c, const TfP& w, const TfSave& sv, const void* d_pred, const TfTmp& tb) {
  const int D = w.dim, Dp = padk(D);
  const size_t MD = (size_t)c.M * Dp;
  WgTap tap{sv.xnp, Dp, 0};
  DN_TRY(weight_grad(c, &tap, 1, D, d_pred, Dp, D, c.G(w.pred_W)));
  DN_TRY(linear_dgrad(c, d_pred, Dp, Dp, c.Wt(w.t_pred), tb.d_xn, Dp, c.dtype));
  return dn_rmsnorm_backward(sv.x + w.depth * MD, Dp, tb.d_xn, Dp, c.dtype, c.B, c.T, D, c.P(w.pred_gamma), nullptr, 0, 0, nullptr, tb.dx,
                             tb.dx_act, c.dtype, Dp, c.frozen ? nullptr : c.G(w.pred_gamma), nullptr, 0, c.red_scratch, c.s);
}

// one layer: tb.dx / tb.dx_act hold d x[l+1] on entry, d x[l] on exit
int tf_backward_layer(const Ctx& c, const TfP& w, int l, const int32_t* lengths, const TfSave& sv, const TfTmp& tb, const float* gb = nullptr,
                      int gb_ld = 0, float* d_gb = nullptr) {
  // An early error return must not leave the side stream running behind the caller's back (it still writes wg_scratch and
  // gradients the caller may reuse or free): unless the layer ends normally -- where the last weight gradient is waited for --
  // the main stream is made to wait for everything the side stream has been given.
  struct JoinOnError {
    const Ctx& c;
    bool armed = true;
    ~JoinOnError() {
      if (armed && c.side && hipEventRecord(c.side->all, c.side->s) == hipSuccess) (void)hipStreamWaitEvent(c.s, c.side->all, 0);
    }
  } join{c};
  const int dtype = c.dtype, es = c.es, B = c.B, T = c.T, M = c.M;
  const int D = w.dim, Dp = padk(D), Dn = padn(D), hd = w.heads * w.dim_head, ip = padk(w.inner), in_n = padn(w.inner);
  const size_t MD = (size_t)M * Dp;
  const float* x = sv.x + l * MD;
  const float* xmid = sv.xmid + l * MD;
  const void* xn1 = eoff(sv.xn1, l * MD, es);
  const void* qkv = eoff(sv.qkv, (size_t)l * M * 3 * hd, es);
  const void* ao = eoff(sv.ao, (size_t)l * M * hd, es);
  const void* xn2 = eoff(sv.xn2, l * MD, es);
  const void* pre = eoff(sv.pre, (size_t)l * M * 2 * ip, es);
  const void* gg = eoff(sv.gg, (size_t)l * M * ip, es);
  const void* fc = eoff(sv.fc, (size_t)l * M * ip, es);
  const bool ada = gb && w.cond_col >= 0;
  const size_t col1 = ada ? w.cond_col + (size_t)(2 * l) * 2 * Dp : 0, col2 = col1 + 2 * Dp;
  // weight gradients on the side stream (WgSide): their operands are this layer's temporaries, so the chain waits for
  //   h_ffout before ff_norm's backward rewrites tb.dx_act, h_out before attn_norm's backward does, and for the last one (the side
  //   stream runs in order) at the end of the layer -- the next layer rewrites tb.d_fc / d_pre / d_qkv, and the caller hands this
  //   layer's gradient range to the all-reduce.
  int h_ffout = -1, h_out = -1, h_last = -1;
  {  // Linear(inner -> D) (:902)
    WgTap tap{fc, ip, 0};
    DN_TRY(weight_grad(c, &tap, 1, w.inner, tb.dx_act, Dp, D, c.G(w.ffout_W(l)), 0, &h_ffout));
    DN_TRY(bias_grad(c, tb.dx, Dp, DN_F32, 1, M, Dp, c.G(w.ffout_b(l)), 0));
    DN_TRY(linear_dgrad(c, tb.dx_act, Dp, Dp, eoff(c.Wt(w.t_ffout), (size_t)l * padn(ip) * Dp, es), tb.d_fc, ip, dtype));
  }
  {  // CausalConv1d(inner, inner, 3) (:894)
    WgTap taps[3];
    for (int j = 0; j < 3; ++j) taps[j] = WgTap{gg, ip, 2 - j};
    DN_TRY(weight_grad(c, taps, 3, w.inner, tb.d_fc, ip, w.inner, c.G(w.ffconv_W(l)), DN_TAG_FFN_CONV_WGRAD, &h_last));
    DN_TRY(bias_grad(c, tb.d_fc, ip, dtype, 1, M, ip, c.G(w.ffconv_b(l)), 0));
    DnGemmParams p = gemm_base(dtype, M, ip, ip, T);
    p.n_terms = 3;
    for (int j = 0; j < 3; ++j) {
      p.terms[j].A = tb.d_fc; p.terms[j].lda = ip; p.terms[j].shift = -(2 - j);
      p.terms[j].W = eoff(c.Wt(w.t_ffconv), ((size_t)l * 3 + j) * padn(ip) * ip, es);
    }
    p.out = tb.d_gg; p.ldo = ip;
    DN_TRY(dn_conv_gemm(&p, c.s));
  }
  DN_TRY(dn_geglu_backward(tb.d_gg, pre, tb.d_pre, dtype, M, ip, c.s));  // (:881-884)
  {  // Linear(D -> 2*inner) (:899), packed columns
    WgTap tap{xn2, Dp, 0};
    DN_TRY(weight_grad(c, &tap, 1, D, tb.d_pre, 2 * ip, 2 * ip, c.G(w.ffin_W(l)), 0, &h_last));
    DN_TRY(bias_grad(c, tb.d_pre, 2 * ip, dtype, 1, M, 2 * ip, c.G(w.ffin_b(l)), 0));
    DN_TRY(linear_dgrad(c, tb.d_pre, 2 * ip, 2 * ip, eoff(c.Wt(w.t_ffin), (size_t)l * padn(Dp) * 2 * ip, es), tb.d_xn, Dp, dtype));
  }
  // ff_norm (:703): d xmid = d x[l+1] + norm'(xmid) d xn2
  DN_TRY(wg_wait(c, h_ffout));
  DN_TRY(dn_rmsnorm_backward(xmid, Dp, tb.d_xn, Dp, dtype, B, T, D, ada ? nullptr : c.P(w.g2(l)), ada ? gb + col2 : nullptr, gb_ld, Dp, tb.dx,
                             tb.dx, tb.dx_act, dtype, Dp, (ada || c.frozen) ? nullptr : c.G(w.g2(l)), ada ? d_gb + col2 : nullptr, gb_ld,
                             c.red_scratch, c.s));
  {  // to_out (:932)
    WgTap tap{ao, hd, 0};
    DN_TRY(weight_grad(c, &tap, 1, hd, tb.dx_act, Dp, D, c.G(w.out_W(l)), 0, &h_out));
    DN_TRY(linear_dgrad(c, tb.dx_act, Dp, Dp, eoff(c.Wt(w.t_out), (size_t)l * padn(hd) * Dp, es), tb.d_ao, hd, dtype));
  }
  {  // Attend (:299-343)
    DnAttnBwdParams a;
    memset(&a, 0, sizeof(a));
    a.q = qkv; a.k = eoff(qkv, hd, es); a.v = eoff(qkv, 2 * hd, es); a.out = ao; a.dout = tb.d_ao;
    a.dq = tb.d_qkv; a.dk = eoff(tb.d_qkv, hd, es); a.dv = eoff(tb.d_qkv, 2 * hd, es);
    a.ldq = a.ldk = a.ldv = 3 * hd; a.ldo = hd; a.lddo = hd; a.lddq = a.lddk = a.lddv = 3 * hd;
    a.B = B; a.T = T; a.heads = w.heads; a.dim_head = w.dim_head; a.dtype = dtype; a.lengths = lengths;
    a.scale = 1.0f / sqrtf((float)w.dim_head);
    a.lse = sv.lse + (size_t)l * B * w.heads * T; a.delta = tb.delta;
    a.dropout_p = c.attn_dropout; a.seed_lo = c.seed_lo; a.seed_hi = c.seed_hi + (uint32_t)l;
    DN_TRY(dn_attention_backward(&a, c.s));
  }
  {  // to_q ; to_kv (:930-931)
    WgTap tap{xn1, Dp, 0};
    DN_TRY(weight_grad(c, &tap, 1, D, tb.d_qkv, 3 * hd, 3 * hd, c.G(w.qkv_W(l)), 0, &h_last));
    DN_TRY(linear_dgrad(c, tb.d_qkv, 3 * hd, 3 * hd, eoff(c.Wt(w.t_qkv), (size_t)l * padn(Dp) * 3 * hd, es), tb.d_xn, Dp, dtype));
  }
  // attn_norm (:691)
  DN_TRY(wg_wait(c, h_out));
  DN_TRY(dn_rmsnorm_backward(x, Dp, tb.d_xn, Dp, dtype, B, T, D, ada ? nullptr : c.P(w.g1(l)), ada ? gb + col1 : nullptr, gb_ld, Dp, tb.dx, tb.dx,
                             tb.dx_act, dtype, Dp, (ada || c.frozen) ? nullptr : c.G(w.g1(l)), ada ? d_gb + col1 : nullptr, gb_ld, c.red_scratch,
                             c.s));
  const int rc = wg_wait(c, h_last);
  join.armed = rc != DN_OK;
  return rc;
}

// ------------------------------------------------------------------------------------------ workspace plan
struct VaePlan {
  void* feat_act;
  void* enc_mid[4];   // encoder WaveNet outputs that feed the next one (arithmetic dtype)
  WaveSave enc[4], dec[4];
  float *params, *z, *kl_rows;
  void* z_act;
  void* dec_mid[4];
  TfSave tf;
  float *rec, *logits, *lsce_rows, *sq_rows, *sums;
  void *rec_act, *dlogits;
  // backward
  WaveTmp wt;
  TfTmp tt;
  float* d_rec;
  void *d_rec_act, *d_mid0, *d_mid1, *d_params;
  float* dz;
  void* wg_scratch;
  size_t wg_scratch_bytes = 0;
  float* red_scratch;
};

size_t max_wgrad_bytes(const DnVaeTrain* m, int B, int T) {
  const int es = esize(m->cfg.dtype);
  size_t mx = 0;
  auto upd = [&](int cin, int cout, int taps, int max_shift) {
    const size_t b = plan_wgrad(cin, cout, taps, max_shift, B, T, es).total();
    mx = b > mx ? b : mx;
  };
  for (int n = 0; n < m->n_wave; ++n)
    for (const WaveP* w : {&m->enc[n], &m->dec[n]}) {
      upd(w->cin, w->cout, 3, 2);
      upd(w->cout, w->cout, 3, 2 << (w->L - 1));
      upd(w->cout, w->cout, w->L, 0);
      const size_t grouped = (size_t)w->L * 2 * w->cout * 3 * padn(w->cout) * 4;  // the stack's blocks as groups of one launch, two slices of partial sums
      mx = grouped > mx ? grouped : mx;
    }
  const TfP& t = m->tf;
  const int hd = t.heads * t.dim_head, ip = padk(t.inner);
  upd(t.dim, 3 * hd, 1, 0); upd(hd, t.dim, 1, 0); upd(t.dim, 2 * ip, 1, 0); upd(t.inner, t.inner, 3, 2); upd(t.inner, t.dim, 1, 0);
  upd(t.dim, t.dim, 1, 0); upd(t.dim, m->cfg.vocab, 1, 0);
  return mx;
}

VaePlan plan_vae_train(const DnVaeTrain* m, int B, int T, Arena& ar, bool need_enc = true) {
  const int es = esize(m->cfg.dtype), D = m->cfg.dim, Dp = padk(D), z = m->cfg.z, V = m->cfg.vocab;
  const size_t M = (size_t)B * T;
  VaePlan p;
  memset(&p, 0, sizeof(p));
  p.feat_act = ar.take(M * Dp * es);
  int widest = 0;
  for (int n = 0; n < m->n_wave; ++n) {
    widest = widest > m->enc[n].cout ? widest : m->enc[n].cout;
    widest = widest > m->dec[n].cout ? widest : m->dec[n].cout;
    if (need_enc) {
      p.enc[n] = plan_wave_save(m->enc[n], (int)M, es, ar);
      p.enc_mid[n] = ar.take(M * padk(m->enc[n].cout) * es);
    }
    p.dec[n] = plan_wave_save(m->dec[n], (int)M, es, ar);
    p.dec_mid[n] = ar.take(M * padk(m->dec[n].cout) * es);
  }
  p.params = (float*)ar.take(M * 2 * z * 4);
  p.z = (float*)ar.take(M * padk(z) * 4);
  p.z_act = ar.take(M * padk(z) * es);
  p.kl_rows = (float*)ar.take(M * 4);
  p.tf = plan_tf_save(m->tf, B, T, es, ar);
  p.rec = (float*)ar.take(M * Dp * 4);
  p.rec_act = ar.take(M * Dp * es);
  p.logits = (float*)ar.take(M * V * 4);
  p.lsce_rows = (float*)ar.take(M * 4 * 4);
  p.sq_rows = (float*)ar.take(M * 4);
  p.sums = (float*)ar.take(64 * 4);
  p.dlogits = ar.take(M * padn(V) * es);
  const size_t wide = M * padk(widest) * es;
  int Lmax = 1;
  for (int n = 0; n < m->n_wave; ++n) Lmax = Lmax > m->enc[n].L ? Lmax : m->enc[n].L;
  p.wt.d_sk = ar.take(wide); p.wt.d_out0 = ar.take(wide * Lmax); p.wt.d_out1 = ar.take(wide * Lmax); p.wt.d_h = ar.take(wide * Lmax);
  p.wt.d_h0 = ar.take(wide);
  p.tt = plan_tf_tmp(m->tf, B, T, es, ar);
  p.d_rec = (float*)ar.take(M * Dp * 4);
  p.d_rec_act = ar.take(M * Dp * es);
  p.d_mid0 = ar.take(wide); p.d_mid1 = ar.take(wide);
  p.d_params = ar.take(M * padk(2 * z) * es);
  p.dz = (float*)ar.take(M * padk(z) * 4);
  p.wg_scratch_bytes = max_wgrad_bytes(m, B, T);
  p.wg_scratch = ar.take(p.wg_scratch_bytes);
  p.red_scratch = (float*)ar.take(((size_t)1024 * 1024 + 4096) * 4 + dn_rmsnorm_backward_scratch_bytes(B, T, D));
  return p;
}

Ctx make_ctx(const DnVaeTrain* m, int B, int T, const VaePlan& pl, hipStream_t s, bool frozen = false) {
  Ctx c;
  c.master = m->master; c.work = m->work; c.aux = m->aux; c.grads = m->grads; c.n_trans = m->n_trans; c.frozen = frozen;
  c.dtype = m->cfg.dtype; c.es = esize(c.dtype); c.B = B; c.T = T; c.M = B * T; c.s = s;
  c.wg_scratch = pl.wg_scratch; c.wg_scratch_bytes = pl.wg_scratch_bytes; c.red_scratch = pl.red_scratch;
  c.side = frozen ? nullptr : wg_side();
  return c;
}

int check_batch(const DnVaeTrain* m, const DnVaeTrainBatch* b, void* ws, size_t ws_bytes, VaePlan* pl, const char* who) {
  DN_CHECK_ARG(m && b && ws, "%s: null argument", who);
  DN_CHECK_ARG(m->master && m->work && m->aux && m->grads, "%s: dn_vae_train_bind has not been called", who);
  DN_CHECK_ARG(b->feat && b->units && b->lengths && b->noise && b->stats, "%s: null batch tensor", who);
  DN_CHECK_ARG(b->B > 0 && b->T > 2 && b->ntokens > 0, "%s: B=%d T=%d ntokens=%d", who, b->B, b->T, b->ntokens);
  DN_CHECK_ARG(((uintptr_t)ws & 255) == 0, "%s: workspace must be 256-byte aligned", who);
  Arena ar{(char*)ws, 0, ws_bytes};
  *pl = plan_vae_train(m, b->B, b->T, ar);
  if (ar.off > ws_bytes) {
    dn_set_error("%s: workspace %zu < required %zu (dn_vae_train_workspace_bytes)", who, ws_bytes, ar.off);
    return DN_EWORKSPACE;
  }
  return DN_OK;
}

// decode_feature (:1109-1116) from pl.z_act: decoder WaveNets -> transformer -> pl.rec (fp32) / pl.rec_act -> decoder_lm -> pl.logits
int vae_decoder_forward(const Ctx& c, const DnVaeTrain* m, const VaePlan& pl, const int32_t* lengths) {
  const int dtype = c.dtype, M = c.M, T = c.T, D = m->cfg.dim, Dp = padk(D), V = m->cfg.vocab;
  const void* cur = pl.z_act;
  for (int n = 0; n < m->n_wave; ++n) {  // decoder WaveNets (:1130-1131); the last opens the fp32 residual stream
    const WaveP& w = m->dec[n];
    const bool lastw = n == m->n_wave - 1;
    DnGemmParams fin = gemm_base(dtype, M, padk(w.cout), padk(w.cout), T);
    if (lastw) {
      fin.out = pl.tf.x; fin.ldo = Dp; fin.out_dtype = DN_F32;
    } else {
      fin.out = pl.dec_mid[n]; fin.ldo = padk(w.cout); fin.out_dtype = dtype;
    }
    DN_TRY(wave_forward(c, w, cur, pl.dec[n], fin));
    cur = pl.dec_mid[n];
  }
  DN_TRY(tf_forward(c, m->tf, lengths, pl.tf, pl.rec, Dp));  // decoded_feature (:1133)
  DN_TRY(dn_convert_rows(pl.rec, DN_F32, Dp, pl.rec_act, dtype, Dp, M, D, c.s));
  DnGemmParams p = gemm_base(dtype, M, V, Dp, T);  // decoder_lm (:1141)
  p.terms[0].A = pl.rec_act; p.terms[0].lda = Dp; p.terms[0].W = c.W(m->lm_W);
  p.bias = c.P(m->lm_b); p.out = pl.logits; p.ldo = V; p.out_dtype = DN_F32;
  return dn_conv_gemm(&p, c.s);
}

// Decoder half of the backward: stage 0 = d logits (pl.dlogits) and the masked-MSE gradient (scale g_mse) through decoder_lm and
// to_pred, 1 .. depth = transformer layers, depth + 1 = decoder WaveNets, leaving d z (fp32 [M, padk(z)]) in pl.dz.
int vae_decoder_backward(const Ctx& c, const DnVaeTrain* m, const VaePlan& pl, const float* feat, const int32_t* lengths, float g_mse,
                         int stage) {
  const int dtype = c.dtype, M = c.M, T = c.T, D = m->cfg.dim, Dp = padk(D), V = m->cfg.vocab, depth = m->tf.depth;
  if (stage == 0) {
    WgTap tap{pl.rec_act, Dp, 0};
    DN_TRY(weight_grad(c, &tap, 1, D, pl.dlogits, padn(V), V, c.G(m->lm_W)));
    DN_TRY(bias_grad(c, pl.dlogits, padn(V), dtype, 1, M, V, c.G(m->lm_b), 0));
    DN_TRY(linear_dgrad(c, pl.dlogits, padn(V), padk(V), c.Wt(m->t_lm), pl.d_rec, Dp, DN_F32));
    DN_TRY(dn_masked_mse_grad(pl.rec, Dp, feat, D, M, D, T, lengths, g_mse, nullptr, pl.d_rec, Dp, 1, pl.d_rec_act, dtype, Dp, c.s));
    return tf_backward_head(c, m->tf, pl.tf, pl.d_rec_act, pl.tt);
  }
  if (stage <= depth) return tf_backward_layer(c, m->tf, depth - stage, lengths, pl.tf, pl.tt);
  const void* d_y = pl.tt.dx_act;  // gradient of the residual stream's first state
  void* mids[2] = {pl.d_mid0, pl.d_mid1};
  for (int n = m->n_wave - 1; n >= 0; --n) {
    const WaveP& w = m->dec[n];
    const void* in = n == 0 ? pl.z_act : pl.dec_mid[n - 1];
    void* d_x = n == 0 ? (void*)pl.dz : mids[n & 1];
    DN_TRY(wave_backward(c, w, in, pl.dec[n], d_y, d_x, n == 0 ? DN_F32 : dtype, pl.wt));
    d_y = d_x;
  }
  return DN_OK;
}

}  // namespace

// =========================================================================================== C ABI
extern "C" int dn_vae_train_create(const DnVaeConfig* cfg, DnVaeTrain** out) {
  DN_CHECK_ARG(cfg && out, "dn_vae_train_create: null argument");
  DN_CHECK_ARG(cfg->n_mults >= 1 && cfg->n_mults <= 4, "dn_vae_train_create: n_mults=%d", cfg->n_mults);
  DN_CHECK_ARG(cfg->dtype == DN_F32 || cfg->dtype == DN_BF16, "dn_vae_train_create: bad dtype");
  DN_CHECK_ARG(cfg->dim % 8 == 0 && (cfg->heads * cfg->dim_head) % 64 == 0 && cfg->z % 4 == 0 && cfg->vocab % 4 == 0 && cfg->vocab <= 1024,
               "dn_vae_train_create: dim %% 8, heads*dim_head %% 64, z %% 4, vocab %% 4 (<= 1024) required");
  DN_CHECK_ARG(cfg->layers >= 1 && cfg->layers <= DN_MAX_TERMS && cfg->stacks >= 1, "dn_vae_train_create: stacks/layers");
  DnVaeTrain* m = new (std::nothrow) DnVaeTrain();
  DN_CHECK_ARG(m != nullptr, "dn_vae_train_create: out of host memory");
  memset(m, 0, sizeof(*m));
  m->cfg = *cfg;
  m->n_wave = cfg->n_mults;
  int64_t cur = 0, tcur = 0, fcur = 0;
  int width = cfg->dim;
  for (int n = 0; n < m->n_wave; ++n) {  // latent_module.py:1053-1065
    WaveP& w = m->enc[n];
    w.cin = width; w.cout = width / cfg->mults[n]; w.S = cfg->stacks; w.L = cfg->layers; w.film_col = -1;
    width = w.cout;
    layout_wave(w, cur, tcur, fcur);
  }
  if (width != 2 * cfg->z) {
    delete m;
    dn_set_error("dn_vae_train_create: encoder width %d != 2*z (%d)", width, 2 * cfg->z);
    return DN_EINVAL;
  }
  for (int n = 0; n < m->n_wave; ++n) {  // :1067-1081
    WaveP& w = m->dec[n];
    const int mult = cfg->mults[m->n_wave - 1 - n];
    w.cout = width * mult; w.cin = n == 0 ? width / 2 : width; w.S = cfg->stacks; w.L = cfg->layers; w.film_col = -1;
    width = w.cout;
    layout_wave(w, cur, tcur, fcur);
  }
  if (width != cfg->dim) {
    delete m;
    dn_set_error("dn_vae_train_create: decoder width %d != dim %d", width, cfg->dim);
    return DN_EINVAL;
  }
  m->tf.dim = cfg->dim; m->tf.depth = cfg->depth; m->tf.heads = cfg->heads; m->tf.dim_head = cfg->dim_head;
  m->tf.inner = (int)((double)cfg->dim * 4 * 2 / 3);
  m->tf.cond_col = -1;
  layout_tf(m->tf, cur, tcur);
  const int64_t Dp = padk(cfg->dim), Vn = padn(cfg->vocab);
  m->lm_W = take(cur, Vn * Dp); m->lm_b = take(cur, Vn);
  m->t_lm = take(tcur, (int64_t)padn(Dp) * padk(cfg->vocab));
  m->n_params = cur; m->n_trans = tcur; m->n_fderived = fcur;
  *out = m;
  return DN_OK;
}

extern "C" void dn_vae_train_destroy(DnVaeTrain* m) { delete m; }

extern "C" int64_t dn_vae_train_param_count(const DnVaeTrain* m) { return m ? m->n_params : 0; }

extern "C" size_t dn_vae_train_aux_bytes(const DnVaeTrain* m) {
  return m ? (size_t)r64(m->n_trans) * esize(m->cfg.dtype) + (size_t)m->n_fderived * 4 + 256 : 0;
}

// offsets (elements) of the packed tensors inside the flat buffers, in the order of diffnorm_amd/packing.py::pack_vae_train:
// per WaveNet (encoders, then decoders) init_W, init_b, conv_W, conv_b, res_W, res_b, skip_W, skip_b[L], final_W, final_b;
// per transformer layer qkv_W, out_W, ffin_W, ffin_b, ffconv_W, ffconv_b, ffout_W, ffout_b, g1, g2; then pred_gamma, pred_W;
// lm_W, lm_b.
extern "C" int dn_vae_train_offsets(const DnVaeTrain* m, int64_t* offsets, int32_t capacity) {
  DN_CHECK_ARG(m && offsets, "dn_vae_train_offsets: null argument");
  const int n = 2 * m->n_wave * 10 + 10 * m->tf.depth + 2 + 2;
  DN_CHECK_ARG(capacity >= n, "dn_vae_train_offsets: capacity %d < %d", capacity, n);
  int k = 0;
  auto wave = [&](const WaveP& w) {
    for (int64_t o : {w.init_W, w.init_b, w.conv_W, w.conv_b, w.res_W, w.res_b, w.skip_W, w.skip_b, w.final_W, w.final_b}) offsets[k++] = o;
  };
  for (int i = 0; i < m->n_wave; ++i) wave(m->enc[i]);
  for (int i = 0; i < m->n_wave; ++i) wave(m->dec[i]);
  const TfP& t = m->tf;
  for (int l = 0; l < t.depth; ++l)
    for (int64_t o : {t.qkv_W(l), t.out_W(l), t.ffin_W(l), t.ffin_b(l), t.ffconv_W(l), t.ffconv_b(l), t.ffout_W(l), t.ffout_b(l), t.g1(l),
                      t.g2(l)})
      offsets[k++] = o;
  offsets[k++] = t.pred_gamma; offsets[k++] = t.pred_W;
  offsets[k++] = m->lm_W; offsets[k++] = m->lm_b;
  return n;
}

// Range of the flat gradient buffer that backward stage `stage` completes (stages as in dn_vae_train_backward).
extern "C" int dn_vae_train_stage_range(const DnVaeTrain* m, int32_t stage, int64_t* offset, int64_t* count) {
  DN_CHECK_ARG(m && offset && count, "dn_vae_train_stage_range: null argument");
  const int depth = m->tf.depth;
  DN_CHECK_ARG(stage >= 0 && stage <= depth + 2, "dn_vae_train_stage_range: stage %d", stage);
  int64_t lo, hi;
  if (stage == 0) {
    lo = m->tf.pred_gamma; hi = m->n_params;
  } else if (stage <= depth) {
    lo = m->tf.layer0 + (int64_t)(depth - stage) * m->tf.layer_stride; hi = lo + m->tf.layer_stride;
  } else if (stage == depth + 1) {
    lo = m->dec[0].init_W; hi = m->tf.layer0;
  } else {
    lo = 0; hi = m->dec[0].init_W;
  }
  *offset = lo; *count = hi - lo;
  return DN_OK;
}

extern "C" int dn_vae_train_bind(DnVaeTrain* m, float* master, void* work, void* aux, float* grads) {
  DN_CHECK_ARG(m && master && work && aux && grads, "dn_vae_train_bind: null argument");
  for (const void* p : {(const void*)master, (const void*)work, (const void*)aux, (const void*)grads})
    DN_CHECK_ARG(((uintptr_t)p & 255) == 0, "dn_vae_train_bind: buffers must be 256-byte aligned");
  DN_CHECK_ARG(m->cfg.dtype == DN_BF16 || (const void*)work == (const void*)master, "dn_vae_train_bind: in f32 mode work must be master");
  m->master = master; m->work = work; m->aux = (char*)aux; m->grads = grads;
  return DN_OK;
}

// aux <- work: the transposed matrices of the data-gradient contractions and the summed skip biases.  Call after every update.
extern "C" int dn_vae_train_refresh(DnVaeTrain* m, void* stream) {
  DN_CHECK_ARG(m && m->work && m->aux, "dn_vae_train_refresh: not bound");
  const int dtype = m->cfg.dtype, es = esize(dtype);
  hipStream_t s = (hipStream_t)stream;
  // matrix [Np][Kp] (Np = padn(N)) -> [padn(Kp)][padk(N)]
  auto tr_strided = [&](int64_t off, int64_t src_stride, int64_t toff, int count, int N, int Kp) -> int {
    return dn_transpose_weights(static_cast<const char*>(m->work) + off * es, dtype, count, src_stride, padk(N), Kp, m->aux + toff * es,
                                (int64_t)padn(Kp) * padk(N), padk(N), padn(Kp), s);
  };
  auto tr = [&](int64_t off, int64_t toff, int count, int N, int Kp) -> int { return tr_strided(off, (int64_t)padn(N) * Kp, toff, count, N, Kp); };
  float* fder = reinterpret_cast<float*>(m->aux + r64(m->n_trans) * es);
  for (int n = 0; n < 2 * m->n_wave; ++n) {
    const WaveP& w = n < m->n_wave ? m->enc[n] : m->dec[n - m->n_wave];
    const int cinp = padk(w.cin), cp = padk(w.cout);
    DN_TRY(tr(w.init_W, w.t_init, 3, w.cout, cinp));
    DN_TRY(tr(w.conv_W, w.t_conv, w.S * w.L * 3, w.cout, cp));
    DN_TRY(tr(w.res_W, w.t_res, w.S * w.L, w.cout, cp));
    DN_TRY(tr(w.skip_W, w.t_skip, w.L, w.cout, cp));
    DN_TRY(tr(w.final_W, w.t_final, 1, w.cout, cp));
    DN_TRY(dn_sum_groups(m->master + w.skip_b, cp, w.L, fder + w.skip_bsum, DN_F32, cp, s));
  }
  const TfP& t = m->tf;
  const int Dp = padk(t.dim), hd = t.heads * t.dim_head, ip = padk(t.inner);
  DN_TRY(tr_strided(t.qkv_W(0), t.layer_stride, t.t_qkv, t.depth, 3 * hd, Dp));
  DN_TRY(tr_strided(t.out_W(0), t.layer_stride, t.t_out, t.depth, t.dim, hd));
  DN_TRY(tr_strided(t.ffin_W(0), t.layer_stride, t.t_ffin, t.depth, 2 * ip, Dp));
  for (int l = 0; l < t.depth; ++l)
    DN_TRY(tr(t.ffconv_W(l), t.t_ffconv + (int64_t)l * 3 * padn(ip) * ip, 3, t.inner, ip));
  DN_TRY(tr_strided(t.ffout_W(0), t.layer_stride, t.t_ffout, t.depth, t.dim, ip));
  DN_TRY(tr(t.pred_W, t.t_pred, 1, t.dim, Dp));
  return tr(m->lm_W, m->t_lm, 1, m->cfg.vocab, Dp);
}

extern "C" size_t dn_vae_train_workspace_bytes(const DnVaeTrain* m, int32_t B, int32_t T) {
  if (!m || B <= 0 || T <= 0) return 0;
  Arena ar{nullptr, 0, 0};
  (void)plan_vae_train(m, B, T, ar);
  return ar.off + 256;
}

// SpeechVAEEncoderDecoder.forward (:1118-1142) + the criterion's loss terms; keeps every activation in the workspace.
extern "C" int dn_vae_train_forward(DnVaeTrain* m, const DnVaeTrainBatch* b, void* workspace, size_t workspace_bytes, void* stream) {
  VaePlan pl;
  DN_TRY(check_batch(m, b, workspace, workspace_bytes, &pl, "dn_vae_train_forward"));
  hipStream_t s = (hipStream_t)stream;
  Ctx c = make_ctx(m, b->B, b->T, pl, s);
  c.attn_dropout = b->attn_dropout; c.seed_lo = b->dropout_seed_lo; c.seed_hi = b->dropout_seed_hi;
  const int dtype = c.dtype, M = c.M, T = c.T, D = m->cfg.dim, Dp = padk(D), z = m->cfg.z, zp = padk(z), V = m->cfg.vocab;
  DN_TRY(dn_convert_rows(b->feat, DN_F32, D, pl.feat_act, dtype, Dp, M, D, s));
  const void* cur = pl.feat_act;
  for (int n = 0; n < m->n_wave; ++n) {  // encoder WaveNets (:1120-1122)
    const WaveP& w = m->enc[n];
    const bool lastw = n == m->n_wave - 1;
    DnGemmParams fin = gemm_base(dtype, M, lastw ? w.cout : padk(w.cout), padk(w.cout), T);
    if (lastw) {
      fin.out = pl.params; fin.ldo = w.cout; fin.out_dtype = DN_F32;
    } else {
      fin.out = pl.enc_mid[n]; fin.ldo = padk(w.cout); fin.out_dtype = dtype;
    }
    DN_TRY(wave_forward(c, w, cur, pl.enc[n], fin));
    cur = pl.enc_mid[n];
  }
  // posterior sample + KL rows (:1124-1127)
  DN_TRY(dn_posterior_sample(pl.params, 2 * z, b->noise, z, pl.z, pl.z_act, dtype, zp, M, z, T, b->lengths, pl.kl_rows, s));
  DN_TRY(vae_decoder_forward(c, m, pl, b->lengths));
  if (b->logits_out) DN_TRY(dn_convert_rows(pl.logits, DN_F32, V, b->logits_out, DN_F32, V, M, V, s));
  if (b->recon_out) DN_TRY(dn_convert_rows(pl.rec, DN_F32, Dp, b->recon_out, DN_F32, D, M, D, s));
  // losses (speech_vae_decoder_loss.py:60-83): LS-CE rows (+ d logits, kept for the backward), squared error, KL
  const float g_ls = b->loss_scale * b->w_lsce / (float)b->ntokens;
  DN_TRY(dn_lsce_loss_grad(pl.logits, V, b->units, M, V, b->label_smoothing, g_ls, pl.lsce_rows, pl.dlogits, dtype, padn(V), s));
  DN_TRY(dn_masked_mse_grad(pl.rec, Dp, b->feat, D, M, D, T, b->lengths, 0.f, pl.sq_rows, nullptr, 0, 0, nullptr, dtype, 0, s));
  DN_TRY(dn_colsum(pl.lsce_rows, 4, DN_F32, 1, M, 4, pl.sums, 0, 1.0f, 0, pl.red_scratch, s));
  DN_TRY(dn_vec_sum(pl.sq_rows, M, pl.sums + 4, 0, pl.red_scratch, s));
  DN_TRY(dn_vec_sum(pl.kl_rows, M, pl.sums + 5, 0, pl.red_scratch, s));
  const int n_valid = b->ntokens;  // frames counted by the masked MSE = sum of the lengths
  hipLaunchKernelGGL(dn::vae_loss_kernel, dim3(1), dim3(1), 0, s, pl.sums, b->stats, b->w_lsce, b->w_mse, b->w_kl, b->label_smoothing, V,
                     1.0f / (float)b->ntokens, 1.0f / ((float)n_valid * (float)D), 1.0f / ((float)b->B * (float)z * (float)T));
  DN_CHECK_LAUNCH("dn_vae_train_forward");
  return DN_OK;
}

// Backward of the step whose activations dn_vae_train_forward left in the workspace; gradients are ADDED to the bound flat
// gradient buffer.  Stages (so the caller can start the all-reduce of a finished range while the rest still runs):
//   0 = losses' gradients + decoder_lm + to_pred;  1 .. depth = transformer layers depth-1 .. 0;  depth+1 = decoder WaveNets +
//   posterior;  depth+2 = encoder WaveNets.  The gradient ranges finish in reverse parameter order.
extern "C" int dn_vae_train_backward(DnVaeTrain* m, const DnVaeTrainBatch* b, int32_t first_stage, int32_t last_stage, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  VaePlan pl;
  DN_TRY(check_batch(m, b, workspace, workspace_bytes, &pl, "dn_vae_train_backward"));
  hipStream_t s = (hipStream_t)stream;
  Ctx c = make_ctx(m, b->B, b->T, pl, s);
  c.attn_dropout = b->attn_dropout; c.seed_lo = b->dropout_seed_lo; c.seed_hi = b->dropout_seed_hi;
  const int dtype = c.dtype, es = c.es, M = c.M, T = c.T, D = m->cfg.dim, Dp = padk(D), z = m->cfg.z, zp = padk(z), V = m->cfg.vocab;
  const int depth = m->tf.depth;
  DN_CHECK_ARG(first_stage >= 0 && last_stage <= depth + 2 && first_stage <= last_stage, "dn_vae_train_backward: stages [%d, %d]", first_stage,
               last_stage);
  for (int stage = first_stage; stage <= last_stage; ++stage) {
    if (stage <= depth + 1) {
      if (stage == 0 && b->ext_dlogits) DN_TRY(dn_convert_rows(b->ext_dlogits, DN_F32, V, pl.dlogits, dtype, padn(V), M, V, s));
      // + masked MSE gradient (:1135-1138): 2 w / (n_valid * D) * (rec - feat)
      const float g_mse = b->loss_scale * b->w_mse * 2.0f / ((float)b->ntokens * (float)D);
      DN_TRY(vae_decoder_backward(c, m, pl, b->feat, b->lengths, g_mse, stage));
      if (stage <= depth) continue;
      // posterior sample + KL (:1124-1127): kl.mean() over the batch of per-sample means over (z, T)
      const float klw = b->loss_scale * b->w_kl / ((float)b->B * (float)z * (float)T);
      DN_TRY(dn_posterior_backward(pl.params, 2 * z, b->noise, z, pl.dz, zp, pl.d_params, dtype, padk(2 * z), M, z, T, b->lengths, klw, s));
    } else {
      const void* d_y = pl.d_params;
      void* mids[2] = {pl.d_mid0, pl.d_mid1};
      for (int n = m->n_wave - 1; n >= 0; --n) {
        const WaveP& w = m->enc[n];
        const void* in = n == 0 ? pl.feat_act : pl.enc_mid[n - 1];
        void* d_x = n == 0 ? nullptr : mids[n & 1];
        DN_TRY(wave_backward(c, w, in, pl.enc[n], d_y, d_x, dtype, pl.wt));
        d_y = d_x;
      }
    }
  }
  (void)es;
  return DN_OK;
}

// ===========================================================================================================================
// Diffusion training step: LatentDiscreteModel.forward (latent_module.py:1514-1613) -- q-sample at a random t,
// the eps-predictor Model(x_t, t) with FiLM / adaptive-norm time conditioning, the min-SNR-5 weighted masked noise MSE, and
// (multitask) the reconstruction losses of x1_hat pushed through the FROZEN VAE decoder (50 * MSE + LS-CE, divided by the
// number of timesteps); the backward pass flows through the frozen decoder's data gradients into the eps-predictor only
// (diff_discrete.py:79-82).  Flat layout, stages and buffers as for the VAE above; the conditioning path (time MLP and the
// 2 S L + 2 depth FiLM / adaptive-norm projections, 45 % of the parameters) stays fp32 in every mode, like the sampling engine.
namespace dn {

// x1 = z + jitter * beta0; x_t = sa[t_b] x1 + s1[t_b] noise  (:1534-1543) -> xt fp32 [M, zl] and its operand copy [M, zp]
__global__ __launch_bounds__(256) void eps_prep_kernel(const float* __restrict__ z, const float* __restrict__ jitter, const float* __restrict__ noise,
                                                       const int32_t* __restrict__ times, const float* __restrict__ sa, const float* __restrict__ s1,
                                                       float beta0, int M, int T, int zl, int zp, float* __restrict__ xt, void* __restrict__ xt_act,
                                                       int act_dtype) {
  const int64_t n = (int64_t)M * zp;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int m = (int)(i / zp), c = (int)(i - (int64_t)m * zp);
    float v = 0.f;
    if (c < zl) {
      const int tb = times[m / T];
      const int64_t o = (int64_t)m * zl + c;
      const float x1 = __fadd_rn(z[o], __fmul_rn(jitter[o], beta0));
      v = __fadd_rn(__fmul_rn(sa[tb], x1), __fmul_rn(s1[tb], noise[o]));
      xt[o] = v;
    }
    if (act_dtype == DN_BF16)
      reinterpret_cast<uint16_t*>(xt_act)[i] = (uint16_t)(pack_bf16x2(v, 0.f) & 0xffff);
    else
      reinterpret_cast<float*>(xt_act)[i] = v;
  }
}

// After the eps-predictor: rows[m] = {sum_c masked (eps_hat - eps)^2, 0, 0, 0}; d_eps[m, c] = g_noise * w_b * diff (valid frames);
// x1_hat = (x_t - s1 eps_hat) / max(sa, 1e-10) in the arithmetic dtype [M, zp] (:1563-1575)
__global__ __launch_bounds__(256) void eps_post_kernel(const float* __restrict__ eps_hat, const float* __restrict__ noise, const float* __restrict__ xt,
                                                       const int32_t* __restrict__ times, const float* __restrict__ sa, const float* __restrict__ s1,
                                                       const float* __restrict__ snr_w, const int32_t* __restrict__ lengths, float g_noise, int M,
                                                       int T, int zl, int zp, float* __restrict__ rows, float* __restrict__ d_eps,
                                                       void* __restrict__ x1_act, int act_dtype) {
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (m >= M) return;
  const int b = m / T, t = m - b * T;
  const bool valid = t < lengths[b];
  const int tb = times[b];
  const float a = sa[tb], s = s1[tb], w = snr_w[b];
  float sq = 0.f;
  for (int c = lane; c < zp; c += 64) {
    float x1 = 0.f, g = 0.f;
    if (c < zl) {
      const int64_t o = (int64_t)m * zl + c;
      const float e = eps_hat[o];
      const float diff = e - noise[o];
      if (valid) {
        sq += diff * diff;
        g = g_noise * w * diff;
      }
      x1 = __fdiv_rn(__fsub_rn(xt[o], __fmul_rn(s, e)), fmaxf(a, 1e-10f));
    }
    d_eps[(int64_t)m * zp + c] = g;
    if (act_dtype == DN_BF16)
      reinterpret_cast<uint16_t*>(x1_act)[(int64_t)m * zp + c] = (uint16_t)(pack_bf16x2(x1, 0.f) & 0xffff);
    else
      reinterpret_cast<float*>(x1_act)[(int64_t)m * zp + c] = x1;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  if (lane == 0) *reinterpret_cast<float4*>(rows + (int64_t)m * 4) = make_float4(sq, 0.f, 0.f, 0.f);
}

// d eps_hat = d_eps (noise part) - s1 / max(sa, 1e-10) * d x1_hat  -> operand copy [M, zp]
__global__ __launch_bounds__(256) void eps_combine_kernel(const float* __restrict__ d_eps, const float* __restrict__ dx1, const int32_t* __restrict__ times,
                                                          const float* __restrict__ sa, const float* __restrict__ s1, int M, int T, int zl, int zp,
                                                          void* __restrict__ out, int act_dtype) {
  const int64_t n = (int64_t)M * zp;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int m = (int)(i / zp), c = (int)(i - (int64_t)m * zp);
    float v = 0.f;
    if (c < zl) {
      const int tb = times[m / T];
      v = d_eps[i] - (dx1 ? s1[tb] / fmaxf(sa[tb], 1e-10f) * dx1[i] : 0.f);
    }
    if (act_dtype == DN_BF16)
      reinterpret_cast<uint16_t*>(out)[i] = (uint16_t)(pack_bf16x2(v, 0.f) & 0xffff);
    else
      reinterpret_cast<float*>(out)[i] = v;
  }
}

// stats: [0] total_loss, [1] nll_loss (label-smoothed CE / n_units), [2] recon_mse_loss, [3] noise_loss, [4] acc  (:1597-1613)
__global__ void eps_loss_kernel(const float* __restrict__ lsce_sums, const float* __restrict__ sq_recon, const float* __restrict__ noise_b, int noise_ld,
                                const float* __restrict__ snr_w, int B, float inv_tz, float eps, int V, float inv_units, float inv_recon,
                                float recon_w, float inv_steps, int multitask, float* __restrict__ stats) {
  float noise = 0.f;
  for (int b = 0; b < B; ++b) noise += snr_w[b] * noise_b[(int64_t)b * noise_ld] * inv_tz;
  noise /= (float)B;
  const float eps_i = eps / (float)(V - 1);
  const float smooth = ((1.0f - eps - eps_i) * lsce_sums[0] + eps_i * lsce_sums[1]) * inv_units;
  const float rmse = sq_recon[0] * inv_recon;
  stats[0] = multitask ? noise + (recon_w * rmse + smooth) * inv_steps : noise;
  stats[1] = smooth; stats[2] = rmse; stats[3] = noise;
  stats[4] = lsce_sums[3] > 0.f ? lsce_sums[2] / lsce_sums[3] : 0.f;
  stats[5] = lsce_sums[3]; stats[6] = 0.f; stats[7] = 0.f;
}

}  // namespace dn

struct DnEpsTrain {
  DnEpsConfig cfg;
  int C, n_cond;
  int64_t w_freq, tc_W, tc_b, cond_W, cond_b, init_W, init_b, final_W, final_b;
  WaveP wn;
  TfP tf;
  int64_t t_init, t_final;
  int64_t f_condT;  // fp32 transposed conditioning projection [padn(C)][n_cond], in the derived-fp32 region
  int64_t n_params, n_trans, n_fderived;
  float* master;
  void* work;
  char* aux;
  float* grads;
  const float* pos_table;
};

namespace {

struct EpsPlan {
  float *cond, *gb, *d_gb, *d_cond, *ds, *xt, *eps, *d_eps, *rows, *sums, *noise_b, *film_rows;
  float* d_cond_parts;  // split-K partial sums of d cond: [32][B][C]
  size_t d_cond_parts_bytes;
  void *xt_act, *h0, *x1_act, *d_eps_act, *tp_act, *d_tp, *d_h0;
  float* pred;
  WaveSave wv;
  TfSave tf;
  WaveTmp wt;
  TfTmp tt;
  VaePlan vae;
  void* wg_scratch;
  size_t wg_scratch_bytes = 0;
  float* red_scratch;
};

size_t eps_max_wgrad_bytes(const DnEpsTrain* m, int B, int T) {
  const int es = esize(m->cfg.dtype);
  size_t mx = 0;
  auto upd = [&](int cin, int cout, int taps, int max_shift, int b, int t, int e) {
    const size_t v = plan_wgrad(cin, cout, taps, max_shift, b, t, e).total();
    mx = v > mx ? v : mx;
  };
  const int D = m->cfg.dim, hd = m->cfg.heads * m->cfg.dim_head, ip = padk(m->tf.inner);
  upd(m->cfg.latent, D, 1, 0, B, T, es); upd(D, D, 3, 2, B, T, es); upd(D, D, 3, 2 << (m->wn.L - 1), B, T, es); upd(D, D, m->wn.L, 0, B, T, es);
  upd(D, 3 * hd, 1, 0, B, T, es); upd(hd, D, 1, 0, B, T, es); upd(D, 2 * ip, 1, 0, B, T, es); upd(m->tf.inner, m->tf.inner, 3, 2, B, T, es);
  upd(m->tf.inner, D, 1, 0, B, T, es); upd(D, D, 1, 0, B, T, es); upd(D, m->cfg.latent, 1, 0, B, T, es);
  upd(m->C, m->n_cond, 1, 0, 1, B, 4);  // the conditioning projection: the batch is its "frame" axis, fp32
  const size_t grouped = (size_t)m->wn.L * 2 * D * 3 * padn(D) * 4;  // the WaveNet stack's blocks as groups of one launch, two slices
  return grouped > mx ? grouped : mx;
}

EpsPlan plan_eps_train(const DnEpsTrain* m, const DnVaeTrain* vae, int B, int T, Arena& ar) {
  const int es = esize(m->cfg.dtype), D = m->cfg.dim, Dp = padk(D), zl = m->cfg.latent, zp = padk(zl);
  const size_t M = (size_t)B * T;
  EpsPlan p;
  memset(&p, 0, sizeof(p));
  p.cond = (float*)ar.take((size_t)B * m->C * 4);
  p.gb = (float*)ar.take((size_t)B * m->n_cond * 4);
  p.d_gb = (float*)ar.take((size_t)B * m->n_cond * 4);
  p.d_cond = (float*)ar.take((size_t)B * m->C * 4);
  p.d_cond_parts_bytes = (size_t)32 * B * m->C * 4;
  p.d_cond_parts_bytes = std::max(p.d_cond_parts_bytes, dn_rows_times_weight_scratch_bytes(B, m->n_cond, m->C));
  p.d_cond_parts = (float*)ar.take(p.d_cond_parts_bytes);
  p.ds = (float*)ar.take((size_t)B * m->C * 4);
  p.xt = (float*)ar.take(M * zl * 4);
  p.xt_act = ar.take(M * zp * es);
  p.h0 = ar.take(M * Dp * es);
  p.wv = plan_wave_save(m->wn, (int)M, es, ar);
  p.tf = plan_tf_save(m->tf, B, T, es, ar);
  p.pred = (float*)ar.take(M * Dp * 4);
  p.tp_act = ar.take(M * Dp * es);
  p.eps = (float*)ar.take(M * zl * 4);
  p.d_eps = (float*)ar.take(M * zp * 4);
  p.x1_act = ar.take(M * zp * es);
  p.d_eps_act = ar.take(M * zp * es);
  p.rows = (float*)ar.take(M * 4 * 4);
  p.sums = (float*)ar.take(64 * 4);
  p.noise_b = (float*)ar.take((size_t)B * 4 * 4);
  p.film_rows = (float*)ar.take(M * 2 * Dp * 4);
  p.d_tp = ar.take(M * Dp * es);
  p.d_h0 = ar.take(M * Dp * es);
  const size_t wide = M * Dp * es;
  p.wt.d_sk = ar.take(wide); p.wt.d_out0 = ar.take(wide * m->wn.L); p.wt.d_out1 = ar.take(wide * m->wn.L); p.wt.d_h = ar.take(wide * m->wn.L);
  p.wt.d_h0 = ar.take(wide);
  p.tt = plan_tf_tmp(m->tf, B, T, es, ar);
  p.wg_scratch_bytes = eps_max_wgrad_bytes(m, B, T);
  p.wg_scratch = ar.take(p.wg_scratch_bytes);
  p.red_scratch = (float*)ar.take(((size_t)1024 * 1024 + 4096) * 4 + dn_rmsnorm_backward_scratch_bytes(B, T, D));
  if (vae) {
    p.vae = plan_vae_train(vae, B, T, ar, false);
    p.vae.z_act = p.x1_act;  // the decoder's input is x1_hat
  }
  return p;
}

Ctx eps_ctx(const DnEpsTrain* m, int B, int T, const EpsPlan& pl, hipStream_t s) {
  Ctx c;
  c.master = m->master; c.work = m->work; c.aux = m->aux; c.grads = m->grads; c.n_trans = m->n_trans; c.frozen = false;
  c.dtype = m->cfg.dtype; c.es = esize(c.dtype); c.B = B; c.T = T; c.M = B * T; c.s = s;
  c.wg_scratch = pl.wg_scratch; c.wg_scratch_bytes = pl.wg_scratch_bytes; c.red_scratch = pl.red_scratch;
  c.side = wg_side();
  return c;
}

int eps_check(const DnEpsTrain* m, const DnVaeTrain* vae, const DnEpsTrainBatch* b, void* ws, size_t ws_bytes, EpsPlan* pl, const char* who) {
  DN_CHECK_ARG(m && b && ws, "%s: null argument", who);
  DN_CHECK_ARG(m->master && m->work && m->aux && m->grads && m->pos_table, "%s: dn_eps_train_bind has not been called", who);
  DN_CHECK_ARG(b->z && b->jitter && b->true_noise && b->times && b->lengths && b->sqrt_ac && b->sqrt_1mac && b->snr_weight && b->stats,
               "%s: null batch tensor", who);
  DN_CHECK_ARG(b->B > 0 && b->T > 2 && b->T <= m->cfg.max_pos && b->timesteps > 1, "%s: B=%d T=%d", who, b->B, b->T);
  DN_CHECK_ARG(!b->multitask || (vae && vae->work && vae->aux && b->feat && b->units && b->n_units > 0 && b->n_frames > 0),
               "%s: the multitask loss needs the bound frozen VAE, feat and units", who);
  DN_CHECK_ARG(!vae || (vae->cfg.z == m->cfg.latent && vae->cfg.dtype == m->cfg.dtype), "%s: VAE latent width / dtype do not match", who);
  DN_CHECK_ARG(((uintptr_t)ws & 255) == 0, "%s: workspace must be 256-byte aligned", who);
  Arena ar{(char*)ws, 0, ws_bytes};
  *pl = plan_eps_train(m, b->multitask ? vae : nullptr, b->B, b->T, ar);
  if (ar.off > ws_bytes) {
    dn_set_error("%s: workspace %zu < required %zu (dn_eps_train_workspace_bytes)", who, ws_bytes, ar.off);
    return DN_EWORKSPACE;
  }
  return DN_OK;
}

}  // namespace

extern "C" int dn_eps_train_create(const DnEpsConfig* cfg, DnEpsTrain** out) {
  DN_CHECK_ARG(cfg && out, "dn_eps_train_create: null argument");
  DN_CHECK_ARG(cfg->dtype == DN_F32 || cfg->dtype == DN_BF16, "dn_eps_train_create: bad dtype");
  DN_CHECK_ARG(cfg->dim % 8 == 0 && (cfg->heads * cfg->dim_head) % 64 == 0 && cfg->latent % 4 == 0 && (cfg->dim * cfg->cond_mult) % 64 == 0,
               "dn_eps_train_create: dim %% 8, heads*dim_head %% 64, latent %% 4, dim*cond_mult %% 64 required");
  DN_CHECK_ARG(cfg->wn_layers >= 1 && cfg->wn_layers <= DN_MAX_TERMS && cfg->wn_stacks >= 1, "dn_eps_train_create: wavenet stacks/layers");
  DnEpsTrain* m = new (std::nothrow) DnEpsTrain();
  DN_CHECK_ARG(m != nullptr, "dn_eps_train_create: out of host memory");
  memset(m, 0, sizeof(*m));
  m->cfg = *cfg;
  const int64_t D = cfg->dim, Dp = padk(D), Dn = padn(D), zl = cfg->latent, zp = padk(zl);
  m->C = cfg->dim * cfg->cond_mult;
  m->n_cond = (cfg->wn_stacks * cfg->wn_layers + 2 * cfg->depth) * 2 * (int)Dp;
  int64_t cur = 0, tcur = 0, fcur = 0;
  // conditioning path first: its gradients complete last (every FiLM / adaptive-norm consumer must have run)
  m->w_freq = take(cur, D / 2); m->tc_W = take(cur, (int64_t)m->C * (D + 1)); m->tc_b = take(cur, m->C);
  m->cond_W = take(cur, (int64_t)padn(m->n_cond) * m->C); m->cond_b = take(cur, m->n_cond);
  m->init_W = take(cur, Dn * zp); m->init_b = take(cur, Dp);
  m->wn.cin = m->wn.cout = cfg->dim; m->wn.S = cfg->wn_stacks; m->wn.L = cfg->wn_layers; m->wn.film_col = 0;
  layout_wave(m->wn, cur, tcur, fcur);
  m->tf.dim = cfg->dim; m->tf.depth = cfg->depth; m->tf.heads = cfg->heads; m->tf.dim_head = cfg->dim_head;
  m->tf.inner = (int)((double)cfg->dim * 4 * 2 / 3);
  m->tf.cond_col = cfg->wn_stacks * cfg->wn_layers * 2 * (int)Dp;
  layout_tf(m->tf, cur, tcur);
  m->final_W = take(cur, (int64_t)padn(zl) * Dp); m->final_b = take(cur, zp);
  m->t_init = take(tcur, (int64_t)padn(zp) * Dp);
  m->t_final = take(tcur, (int64_t)padn(Dp) * zp);
  m->f_condT = take(fcur, (int64_t)padn(m->C) * m->n_cond);
  m->n_params = cur; m->n_trans = tcur; m->n_fderived = fcur;
  *out = m;
  return DN_OK;
}

extern "C" void dn_eps_train_destroy(DnEpsTrain* m) { delete m; }
extern "C" int64_t dn_eps_train_param_count(const DnEpsTrain* m) { return m ? m->n_params : 0; }
extern "C" size_t dn_eps_train_aux_bytes(const DnEpsTrain* m) {
  return m ? (size_t)r64(m->n_trans) * esize(m->cfg.dtype) + (size_t)m->n_fderived * 4 + 256 : 0;
}

// table order (diffnorm_amd/packing.py::eps_train_entries): w_freq, tc_W, tc_b, cond_W, cond_b, init_W, init_b, the WaveNet's 10
// tensors, per layer qkv_W, out_W, ffin_W, ffin_b, ffconv_W, ffconv_b, ffout_W, ffout_b; pred_gamma, pred_W, final_W, final_b
extern "C" int dn_eps_train_offsets(const DnEpsTrain* m, int64_t* offsets, int32_t capacity) {
  DN_CHECK_ARG(m && offsets, "dn_eps_train_offsets: null argument");
  const int n = 7 + 10 + 8 * m->tf.depth + 2 + 2;
  DN_CHECK_ARG(capacity >= n, "dn_eps_train_offsets: capacity %d < %d", capacity, n);
  int k = 0;
  for (int64_t o : {m->w_freq, m->tc_W, m->tc_b, m->cond_W, m->cond_b, m->init_W, m->init_b}) offsets[k++] = o;
  const WaveP& w = m->wn;
  for (int64_t o : {w.init_W, w.init_b, w.conv_W, w.conv_b, w.res_W, w.res_b, w.skip_W, w.skip_b, w.final_W, w.final_b}) offsets[k++] = o;
  const TfP& t = m->tf;
  for (int l = 0; l < t.depth; ++l)
    for (int64_t o : {t.qkv_W(l), t.out_W(l), t.ffin_W(l), t.ffin_b(l), t.ffconv_W(l), t.ffconv_b(l), t.ffout_W(l), t.ffout_b(l)}) offsets[k++] = o;
  for (int64_t o : {t.pred_gamma, t.pred_W, m->final_W, m->final_b}) offsets[k++] = o;
  return n;
}

// stages: 0 = losses + frozen VAE decoder + final_proj + to_pred; 1 .. depth = layers depth-1 .. 0; depth+1 = WaveNet + init_conv;
// depth+2 = conditioning path (time MLP + FiLM / adaptive-norm projections)
extern "C" int dn_eps_train_stage_range(const DnEpsTrain* m, int32_t stage, int64_t* offset, int64_t* count) {
  DN_CHECK_ARG(m && offset && count, "dn_eps_train_stage_range: null argument");
  const int depth = m->tf.depth;
  DN_CHECK_ARG(stage >= 0 && stage <= depth + 2, "dn_eps_train_stage_range: stage %d", stage);
  int64_t lo, hi;
  if (stage == 0) {
    lo = m->tf.pred_gamma; hi = m->n_params;
  } else if (stage <= depth) {
    lo = m->tf.layer0 + (int64_t)(depth - stage) * m->tf.layer_stride; hi = lo + m->tf.layer_stride;
  } else if (stage == depth + 1) {
    lo = m->init_W; hi = m->tf.layer0;
  } else {
    lo = 0; hi = m->init_W;
  }
  *offset = lo; *count = hi - lo;
  return DN_OK;
}

extern "C" int dn_eps_train_bind(DnEpsTrain* m, float* master, void* work, void* aux, float* grads, const float* pos_table) {
  DN_CHECK_ARG(m && master && work && aux && grads && pos_table, "dn_eps_train_bind: null argument");
  for (const void* p : {(const void*)master, (const void*)work, (const void*)aux, (const void*)grads, (const void*)pos_table})
    DN_CHECK_ARG(((uintptr_t)p & 255) == 0, "dn_eps_train_bind: buffers must be 256-byte aligned");
  DN_CHECK_ARG(m->cfg.dtype == DN_BF16 || (const void*)work == (const void*)master, "dn_eps_train_bind: in f32 mode work must be master");
  m->master = master; m->work = work; m->aux = (char*)aux; m->grads = grads; m->pos_table = pos_table;
  return DN_OK;
}

extern "C" int dn_eps_train_refresh(DnEpsTrain* m, void* stream) {
  DN_CHECK_ARG(m && m->work && m->aux, "dn_eps_train_refresh: not bound");
  const int dtype = m->cfg.dtype, es = esize(dtype);
  hipStream_t s = (hipStream_t)stream;
  auto tr_strided = [&](int64_t off, int64_t src_stride, int64_t toff, int count, int N, int Kp) -> int {
    return dn_transpose_weights(static_cast<const char*>(m->work) + off * es, dtype, count, src_stride, padk(N), Kp, m->aux + toff * es,
                                (int64_t)padn(Kp) * padk(N), padk(N), padn(Kp), s);
  };
  auto tr = [&](int64_t off, int64_t toff, int count, int N, int Kp) -> int { return tr_strided(off, (int64_t)padn(N) * Kp, toff, count, N, Kp); };
  float* fder = reinterpret_cast<float*>(m->aux + r64(m->n_trans) * es);
  const WaveP& w = m->wn;
  const int D = m->cfg.dim, Dp = padk(D), zl = m->cfg.latent, zp = padk(zl);
  DN_TRY(tr(m->init_W, m->t_init, 1, D, zp));
  DN_TRY(tr(w.init_W, w.t_init, 3, D, Dp));
  DN_TRY(tr(w.conv_W, w.t_conv, w.S * w.L * 3, D, Dp));
  DN_TRY(tr(w.res_W, w.t_res, w.S * w.L, D, Dp));
  DN_TRY(tr(w.skip_W, w.t_skip, w.L, D, Dp));
  DN_TRY(tr(w.final_W, w.t_final, 1, D, Dp));
  DN_TRY(dn_sum_groups(m->master + w.skip_b, Dp, w.L, fder + w.skip_bsum, DN_F32, Dp, s));
  const TfP& t = m->tf;
  const int hd = t.heads * t.dim_head, ip = padk(t.inner);
  DN_TRY(tr_strided(t.qkv_W(0), t.layer_stride, t.t_qkv, t.depth, 3 * hd, Dp));
  DN_TRY(tr_strided(t.out_W(0), t.layer_stride, t.t_out, t.depth, D, hd));
  DN_TRY(tr_strided(t.ffin_W(0), t.layer_stride, t.t_ffin, t.depth, 2 * ip, Dp));
  for (int l = 0; l < t.depth; ++l) DN_TRY(tr(t.ffconv_W(l), t.t_ffconv + (int64_t)l * 3 * padn(ip) * ip, 3, t.inner, ip));
  DN_TRY(tr_strided(t.ffout_W(0), t.layer_stride, t.t_ffout, t.depth, D, ip));
  DN_TRY(tr(t.pred_W, t.t_pred, 1, D, Dp));
  DN_TRY(tr(m->final_W, m->t_final, 1, zl, Dp));
  // the fp32 conditioning projection [n_cond][C] -> [padn(C)][n_cond], from the master buffer (option cond_stream = 0 only: by default
  // its data gradient streams the master matrix as it lies)
  if (cond_streamed()) return DN_OK;
  return dn_transpose_weights(m->master + m->cond_W, DN_F32, 1, (int64_t)padn(m->n_cond) * m->C, m->n_cond, m->C, fder + m->f_condT,
                              (int64_t)padn(m->C) * m->n_cond, m->n_cond, padn(m->C), s);
}

extern "C" size_t dn_eps_train_workspace_bytes(const DnEpsTrain* m, const DnVaeTrain* vae, int32_t B, int32_t T) {
  if (!m || B <= 0 || T <= 0) return 0;
  Arena ar{nullptr, 0, 0};
  (void)plan_eps_train(m, vae, B, T, ar);
  return ar.off + 256;
}

extern "C" int dn_eps_train_forward(DnEpsTrain* m, DnVaeTrain* vae, const DnEpsTrainBatch* b, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  EpsPlan pl;
  DN_TRY(eps_check(m, vae, b, workspace, workspace_bytes, &pl, "dn_eps_train_forward"));
  hipStream_t s = (hipStream_t)stream;
  Ctx c = eps_ctx(m, b->B, b->T, pl, s);
  c.attn_dropout = b->attn_dropout; c.seed_lo = b->dropout_seed_lo; c.seed_hi = b->dropout_seed_hi;
  const int dtype = c.dtype, B = b->B, T = b->T, M = c.M, D = m->cfg.dim, Dp = padk(D), zl = m->cfg.latent, zp = padk(zl);
  const int ew = (int)std::min<int64_t>(((int64_t)M * zp + 255) / 256, 4096);
  hipLaunchKernelGGL(dn::eps_prep_kernel, dim3(ew), dim3(256), 0, s, b->z, b->jitter, b->true_noise, b->times, b->sqrt_ac, b->sqrt_1mac, b->beta0, M, T,
                     zl, zp, pl.xt, pl.xt_act, dtype);
  // conditioning: to_time_cond (:741-745) and all FiLM / adaptive-norm projections at once (:507,517,624,637), fp32
  DN_TRY(dn_time_cond(b->times, B, c.P(m->w_freq), D / 2, c.P(m->tc_W), c.P(m->tc_b), m->C, pl.cond, nullptr, DN_F32, m->C, s));
  {
    DnGemmParams p = gemm_base(DN_F32, B, m->n_cond, m->C, 1);
    p.terms[0].A = pl.cond; p.terms[0].lda = m->C; p.terms[0].W = c.P(m->cond_W);
    p.bias = c.P(m->cond_b); p.out = pl.gb; p.ldo = m->n_cond; p.out_dtype = DN_F32;
    DN_TRY(dn_conv_gemm(&p, s));
  }
  {  // init_conv 1x1 (:734,864)
    DnGemmParams p = gemm_base(dtype, M, Dp, zp, T);
    p.terms[0].A = pl.xt_act; p.terms[0].lda = zp; p.terms[0].W = c.W(m->init_W);
    p.bias = c.P(m->init_b); p.out = pl.h0; p.ldo = Dp;
    DN_TRY(dn_conv_gemm(&p, s));
  }
  {  // WaveNet; its final conv adds the positional embedding and opens the fp32 residual stream (:865-868)
    DnGemmParams fin = gemm_base(dtype, M, Dp, Dp, T);
    fin.epilogue = DN_EPI_POSEMB; fin.pos_table = m->pos_table; fin.pos_ld = Dp; fin.lengths = b->lengths;
    fin.out = pl.tf.x; fin.ldo = Dp; fin.out_dtype = DN_F32;
    DN_TRY(wave_forward(c, m->wn, pl.h0, pl.wv, fin, pl.gb, m->n_cond));
  }
  DN_TRY(tf_forward(c, m->tf, b->lengths, pl.tf, pl.pred, Dp, pl.gb, m->n_cond));
  DN_TRY(dn_convert_rows(pl.pred, DN_F32, Dp, pl.tp_act, dtype, Dp, M, D, s));
  {  // final_proj (:807,875)
    DnGemmParams p = gemm_base(dtype, M, zl, Dp, T);
    p.terms[0].A = pl.tp_act; p.terms[0].lda = Dp; p.terms[0].W = c.W(m->final_W);
    p.bias = c.P(m->final_b); p.out = pl.eps; p.ldo = zl; p.out_dtype = DN_F32;
    DN_TRY(dn_conv_gemm(&p, s));
  }
  if (b->eps_out) DN_TRY(dn_convert_rows(pl.eps, DN_F32, zl, b->eps_out, DN_F32, zl, M, zl, s));
  // noise loss (:1563-1571): masked squared error, per-sample mean over ALL T*z, min-SNR weight, batch mean
  const float g_noise = b->loss_scale * 2.0f / ((float)B * (float)T * (float)zl);
  hipLaunchKernelGGL(dn::eps_post_kernel, dim3((M + 3) / 4), dim3(256), 0, s, pl.eps, b->true_noise, pl.xt, b->times, b->sqrt_ac, b->sqrt_1mac,
                     b->snr_weight, b->lengths, g_noise, M, T, zl, zp, pl.rows, pl.d_eps, pl.x1_act, dtype);
  DN_TRY(dn_colsum(pl.rows, 4, DN_F32, B, T, 4, pl.noise_b, 4, 1.0f, 0, pl.red_scratch, s));
  (void)hipMemsetAsync(pl.sums, 0, 64 * 4, s);
  if (b->multitask) {  // x1_hat through the frozen VAE decoder (:1574-1596)
    const Ctx vc = make_ctx(vae, B, T, pl.vae, s, true);
    const int V = vae->cfg.vocab, Dv = vae->cfg.dim;
    DN_TRY(vae_decoder_forward(vc, vae, pl.vae, b->lengths));
    const float g_ls = b->loss_scale / ((float)b->timesteps * (float)b->n_units);
    DN_TRY(dn_lsce_loss_grad(pl.vae.logits, V, b->units, M, V, b->label_smoothing, g_ls, pl.vae.lsce_rows, pl.vae.dlogits, dtype, padn(V), s));
    DN_TRY(dn_masked_mse_grad(pl.vae.rec, padk(Dv), b->feat, Dv, M, Dv, T, b->lengths, 0.f, pl.vae.sq_rows, nullptr, 0, 0, nullptr, dtype, 0, s));
    DN_TRY(dn_colsum(pl.vae.lsce_rows, 4, DN_F32, 1, M, 4, pl.sums, 0, 1.0f, 0, pl.red_scratch, s));
    DN_TRY(dn_vec_sum(pl.vae.sq_rows, M, pl.sums + 4, 0, pl.red_scratch, s));
  }
  const int Vv = vae ? vae->cfg.vocab : 1004, Dv = vae ? vae->cfg.dim : 1;
  hipLaunchKernelGGL(dn::eps_loss_kernel, dim3(1), dim3(1), 0, s, pl.sums, pl.sums + 4, pl.noise_b, 4, b->snr_weight, B, 1.0f / ((float)T * (float)zl),
                     b->label_smoothing, Vv, b->n_units > 0 ? 1.0f / (float)b->n_units : 0.f,
                     b->n_frames > 0 ? 1.0f / ((float)b->n_frames * (float)Dv) : 0.f, b->recon_weight, 1.0f / (float)b->timesteps, b->multitask,
                     b->stats);
  DN_CHECK_LAUNCH("dn_eps_train_forward");
  return DN_OK;
}

extern "C" int dn_eps_train_backward(DnEpsTrain* m, DnVaeTrain* vae, const DnEpsTrainBatch* b, int32_t first_stage, int32_t last_stage,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  EpsPlan pl;
  DN_TRY(eps_check(m, vae, b, workspace, workspace_bytes, &pl, "dn_eps_train_backward"));
  hipStream_t s = (hipStream_t)stream;
  Ctx c = eps_ctx(m, b->B, b->T, pl, s);
  c.attn_dropout = b->attn_dropout; c.seed_lo = b->dropout_seed_lo; c.seed_hi = b->dropout_seed_hi;
  const int dtype = c.dtype, B = b->B, T = b->T, M = c.M, D = m->cfg.dim, Dp = padk(D), zl = m->cfg.latent, zp = padk(zl);
  const int depth = m->tf.depth;
  DN_CHECK_ARG(first_stage >= 0 && last_stage <= depth + 2 && first_stage <= last_stage, "dn_eps_train_backward: stages [%d, %d]", first_stage,
               last_stage);
  const int ew = (int)std::min<int64_t>(((int64_t)M * zp + 255) / 256, 4096);
  for (int stage = first_stage; stage <= last_stage; ++stage) {
    if (stage == 0) {
      (void)hipMemsetAsync(pl.d_gb, 0, (size_t)B * m->n_cond * 4, s);
      const float* dx1 = nullptr;
      if (b->multitask) {  // d total / d x1_hat: the frozen decoder's data gradients (no parameter gradients)
        const Ctx vc = make_ctx(vae, B, T, pl.vae, s, true);
        const float g_mse = b->loss_scale * b->recon_weight * 2.0f / ((float)b->timesteps * (float)b->n_frames * (float)vae->cfg.dim);
        for (int vs = 0; vs <= vae->tf.depth + 1; ++vs) DN_TRY(vae_decoder_backward(vc, vae, pl.vae, b->feat, b->lengths, g_mse, vs));
        dx1 = pl.vae.dz;
      }
      hipLaunchKernelGGL(dn::eps_combine_kernel, dim3(ew), dim3(256), 0, s, pl.d_eps, dx1, b->times, b->sqrt_ac, b->sqrt_1mac, M, T, zl, zp,
                         pl.d_eps_act, dtype);
      // final_proj (:875)
      WgTap tap{pl.tp_act, Dp, 0};
      DN_TRY(weight_grad(c, &tap, 1, D, pl.d_eps_act, zp, zl, c.G(m->final_W)));
      DN_TRY(bias_grad(c, pl.d_eps_act, zp, dtype, 1, M, zp, c.G(m->final_b), 0));
      DN_TRY(linear_dgrad(c, pl.d_eps_act, zp, zp, c.Wt(m->t_final), pl.d_tp, Dp, dtype));
      DN_TRY(tf_backward_head(c, m->tf, pl.tf, pl.d_tp, pl.tt));
    } else if (stage <= depth) {
      DN_TRY(tf_backward_layer(c, m->tf, depth - stage, b->lengths, pl.tf, pl.tt, pl.gb, m->n_cond, pl.d_gb));
    } else if (stage == depth + 1) {
      // the positional embedding is a constant: the residual stream's gradient goes straight into the WaveNet's final conv
      DN_TRY(wave_backward(c, m->wn, pl.h0, pl.wv, pl.tt.dx_act, pl.d_h0, dtype, pl.wt, pl.gb, m->n_cond, pl.d_gb, pl.film_rows));
      WgTap tap{pl.xt_act, zp, 0};  // init_conv 1x1
      DN_TRY(weight_grad(c, &tap, 1, zl, pl.d_h0, Dp, D, c.G(m->init_W)));
      DN_TRY(bias_grad(c, pl.d_h0, Dp, dtype, 1, M, Dp, c.G(m->init_b), 0));
    } else {
      // conditioning path, fp32: gb = cond W_c^T + b_c  ->  d b_c, d W_c = d gb^T cond (the batch is the contracted axis), d cond
      Ctx cf = c;
      cf.dtype = DN_F32; cf.es = 4; cf.B = 1; cf.T = B; cf.M = B;
      cf.work = m->master;  // fp32 operands
      DN_TRY(dn_colsum(pl.d_gb, m->n_cond, DN_F32, 1, B, m->n_cond, c.G(m->cond_b), 0, 1.0f, 1, pl.red_scratch, s));
      WgTap tap{pl.cond, m->C, 0};
      DN_TRY(weight_grad(cf, &tap, 1, m->C, pl.d_gb, m->n_cond, m->n_cond, c.G(m->cond_W)));
      if (cond_streamed()) {
        // d cond = d gb [B, n_cond] . W_c: the fp32 master matrix [n_cond][C] streamed once as it lies (dn_rows_times_weight: no
        // transposed copy per update -- 236 us -- and no split-K contraction over it -- 235 us)
        DN_TRY(dn_rows_times_weight(pl.d_gb, m->n_cond, B, m->master + m->cond_W, m->C, m->n_cond, m->C, pl.d_cond, pl.d_cond_parts, s));
      } else {
        // d cond = d gb [B, n_cond] . W_c: B rows against a 57 k-deep contraction -- 16 output tiles on 256 CUs (measured 7.0 ms = 16 %
        // of a diffusion update when it ran un-split).  Split K over `ks` groups (a K-slice of the packed transpose: ldw = n_cond),
        // partial sums per group, then a fixed-order sum of the groups: every CU gets a slice and the 470 MB of weights stream once.
        float* fder = reinterpret_cast<float*>(m->aux + r64(m->n_trans) * c.es);
        int ks = 1;
        for (int cand = 32; cand >= 2; --cand)
          if (m->n_cond % (cand * 32) == 0 && (size_t)cand * B * m->C * 4 <= pl.d_cond_parts_bytes) { ks = cand; break; }
        DnGemmParams p = gemm_base(DN_F32, B, m->C, m->n_cond / ks, 1);
        p.groups = ks;
        p.terms[0].A = pl.d_gb; p.terms[0].lda = m->n_cond; p.terms[0].a_gstride = m->n_cond / ks;
        p.terms[0].W = fder + m->f_condT; p.terms[0].ldw = m->n_cond; p.terms[0].w_gstride = m->n_cond / ks;
        p.out = ks > 1 ? pl.d_cond_parts : pl.d_cond; p.ldo = m->C; p.out_dtype = DN_F32; p.out_gstride = (int64_t)B * m->C;
        DN_TRY(dn_conv_gemm(&p, s));
        if (ks > 1) DN_TRY(dn_sum_groups(pl.d_cond_parts, (int64_t)B * m->C, ks, pl.d_cond, DN_F32, (int64_t)B * m->C, s));
      }
      DN_TRY(dn_time_cond_backward(b->times, B, c.P(m->w_freq), D / 2, c.P(m->tc_W), c.P(m->tc_b), m->C, pl.d_cond, m->C, pl.ds, c.G(m->w_freq),
                                   c.G(m->tc_W), c.G(m->tc_b), s));
    }
  }
  DN_CHECK_LAUNCH("dn_eps_train_backward");
  return DN_OK;
}
