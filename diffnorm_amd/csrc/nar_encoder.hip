// The speech ENCODER of the NAR S2UT model behind the mask-predict loop: S2STransformerEncoder
// (research/TranSpeech/nar_transformer.py:40-76, no speaker embedding) = fairseq's S2TTransformerEncoder._forward
// (fairseq/models/speech_to_text/s2t_transformer.py:345-373): Conv1dSubsampler (fairseq/models/speech_to_text/modules/convolution.py:
// 13-57: two Conv1d(k, stride 2, padding k/2) + GLU over the channels) -> sqrt(D) scaling -> sinusoidal positions of the padding
// mask -> pre-norm TransformerEncoderLayers (fairseq/modules/transformer_layer.py:163-226: self-attention with the key-padding mask,
// ReLU FFN, all biases) -> LayerNorm.  One pass per utterance batch, in front of the loop: generate() of the research generator
// (research/TranSpeech/iterative_refinement_generator.py:131-160) then runs from [B, L, 80] features on this library alone.
// A strided conv is a contraction over gathered rows: two small gather kernels (the second applies the first conv's GLU on the way)
// feed dn_conv_gemm; the layers are dn_conv_gemm / dn_attention launches and the LayerNorm kernel of nar_decoder.hip.
#include <new>

#include <algorithm>

#include "common.h"
#include "engine.h"
#include "nar_common.h"

using namespace dn;

namespace dn {

__device__ __forceinline__ void store_act(void* p, int64_t off, int dtype, float v) {
  if (dtype == DN_BF16X3) store1_split(p, off, v);
  else if (dn_is16(dtype)) reinterpret_cast<uint16_t*>(p)[off] = to_h16(dtype, v);
  else reinterpret_cast<float*>(p)[off] = v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Rows of the contraction that IS a Conv1d(k, stride 2, padding k / 2) over [B, Lin, C]: cols[b * Lout + t][j * C + c] = in[b][2 t + j - k / 2][c]
// (zero outside [0, Lin); columns >= k * C zero: the K padding).  GLU: `in` is the previous conv's output [B, Lin, 2 C] and the gathered
// value is a * sigmoid(g) with a = in[..][c], g = in[..][C + c] (F.glu over the channels, convolution.py:56).  The whole PADDED batch is
// convolved, frames behind an utterance's end included, as upstream.
template <bool GLU>
__global__ __launch_bounds__(256) void strided_rows_kernel(const float* __restrict__ in, int B, int Lin, int Lout, int C, int k, int Kp,
                                                           void* __restrict__ cols, int dtype) {
  const int64_t n = (int64_t)B * Lout * Kp;
  const int ld_in = GLU ? 2 * C : C;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int col = (int)(i % Kp);
    const int64_t row = i / Kp;
    const int t = (int)(row % Lout), b = (int)(row / Lout);
    float v = 0.f;
    if (col < k * C) {
      const int j = col / C, c = col - j * C;
      const int src = 2 * t + j - k / 2;
      if (src >= 0 && src < Lin) {
        const float* r = in + ((int64_t)b * Lin + src) * ld_in;
        v = GLU ? r[c] * sigmoidf_(r[C + c]) : r[c];
      }
    }
    store_act(cols, i, dtype, v);
  }
}

__device__ __forceinline__ int subsampled_len(int len) { return len <= 0 ? 0 : (len - 1) / 2 + 1; }  // floor((len - 1) / 2 + 1), convolution.py:45-49

// x[b, t, :] = sqrt(D) * glu(y[b, t, :]) + P[pos], pos = pad + 1 + t for t < len2[b] and `pad` (the zero row) behind it: make_positions on the
// padding mask itself (s2t_transformer.py:347-351; fairseq/utils.py:256-266).  Also writes the subsampled lengths.
__global__ __launch_bounds__(256) void enc_embed_kernel(const float* __restrict__ y, const int32_t* __restrict__ src_len, int n_conv, int B, int S, int D,
                                                        const float* __restrict__ pos_table, int pad, float scale, float* __restrict__ x,
                                                        int32_t* __restrict__ out_len) {
  const int64_t n = (int64_t)B * S * D;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int d = (int)(i % D);
    const int64_t row = i / D;
    const int t = (int)(row % S), b = (int)(row / S);
    int len = src_len[b];
    for (int l = 0; l < n_conv; ++l) len = subsampled_len(len);
    if (t == 0 && d == 0) out_len[b] = len < S ? len : S;
    const float* r = y + row * 2 * D;
    const int pos = t < len ? pad + 1 + t : pad;
    x[i] = fmaf(scale, r[d] * sigmoidf_(r[D + d]), pos_table[(int64_t)pos * D + d]);
  }
}
}  // namespace dn

struct DnNarEnc {
  DnNarEncConfig cfg;
  const void *conv0_W, *conv1_W, *qkv_W, *so_W, *fc1_W, *fc2_W;
  const float *conv0_b, *conv1_b, *pos, *qkv_b, *so_b, *fc1_b, *fc2_b, *ln_g, *ln_b, *fin_g, *fin_b;
};
static const int kNarEncTensors = 17;

extern "C" int dn_nar_encoder_create(const DnNarEncConfig* cfg, const void* const* w, int32_t n, DnNarEnc** out) {
  DN_CHECK_ARG(cfg && w && out, "dn_nar_encoder_create: null argument");
  DN_CHECK_ARG(n == kNarEncTensors, "dn_nar_encoder_create: expected %d packed tensors, got %d", kNarEncTensors, n);
  DN_CHECK_ARG(cfg->dtype == DN_F32 || cfg->dtype == DN_BF16 || cfg->dtype == DN_BF16X3 || cfg->dtype == DN_F16, "dn_nar_encoder_create: bad dtype");
  DN_CHECK_ARG(cfg->dim > 0 && cfg->dim % 64 == 0 && cfg->dim <= 1024 && cfg->dim % cfg->heads == 0 && (cfg->dim / cfg->heads) % 4 == 0,
               "dn_nar_encoder_create: embed dim %d must be a multiple of 64 (<= 1024) and of the head count", cfg->dim);
  DN_CHECK_ARG(cfg->ffn % 64 == 0 && cfg->layers >= 1 && cfg->input_dim > 0 && cfg->conv_channels > 0 && cfg->conv_channels % 8 == 0 &&
                   (cfg->kernel * (cfg->conv_channels / 2)) % 64 == 0 && cfg->kernel >= 1 && cfg->kernel % 2 == 1,
               "dn_nar_encoder_create: ffn %% 64, conv_channels %% 8, odd kernel, kernel * conv_channels / 2 a multiple of 64");
  for (int i = 0; i < n; ++i) DN_CHECK_ARG(w[i] != nullptr, "dn_nar_encoder_create: packed tensor %d is null", i);
  DnNarEnc* m = new (std::nothrow) DnNarEnc();
  DN_CHECK_ARG(m != nullptr, "dn_nar_encoder_create: out of host memory");
  m->cfg = *cfg;
  int i = 0;
  m->conv0_W = w[i++]; m->conv0_b = (const float*)w[i++]; m->conv1_W = w[i++]; m->conv1_b = (const float*)w[i++]; m->pos = (const float*)w[i++];
  m->qkv_W = w[i++]; m->qkv_b = (const float*)w[i++]; m->so_W = w[i++]; m->so_b = (const float*)w[i++];
  m->fc1_W = w[i++]; m->fc1_b = (const float*)w[i++]; m->fc2_W = w[i++]; m->fc2_b = (const float*)w[i++];
  m->ln_g = (const float*)w[i++]; m->ln_b = (const float*)w[i++]; m->fin_g = (const float*)w[i++]; m->fin_b = (const float*)w[i++];
  *out = m;
  return DN_OK;
}

extern "C" void dn_nar_encoder_destroy(DnNarEnc* m) { delete m; }

extern "C" int32_t dn_nar_encoder_out_frames(int32_t L) {  // two stride-2 layers with padding k / 2 (any odd k): floor((L - 1) / 2) + 1, twice
  const int s1 = L <= 0 ? 0 : (L - 1) / 2 + 1;
  return s1 <= 0 ? 0 : (s1 - 1) / 2 + 1;
}

namespace {
inline int side_dt(int dtype) { return dtype == DN_BF16X3 ? DN_F32 : dtype; }

struct EncBufs { void *cols0, *cols1, *xn, *qkv, *ao, *h; float *y0, *y1, *x; };
EncBufs plan_enc(const DnNarEnc* m, int B, int L, Arena& ar) {
  const DnNarEncConfig& c = m->cfg;
  const int es = esize(c.dtype), D = c.dim, S1 = L <= 0 ? 0 : (L - 1) / 2 + 1, S = dn_nar_encoder_out_frames(L);
  const int K0 = padk(c.kernel * c.input_dim), K1 = c.kernel * (c.conv_channels / 2);
  EncBufs b;
  b.cols0 = ar.take((size_t)B * S1 * K0 * es);
  b.y0 = (float*)ar.take((size_t)B * S1 * c.conv_channels * 4);
  b.cols1 = ar.take((size_t)B * S * K1 * es);
  b.y1 = (float*)ar.take((size_t)B * S * 2 * D * 4);
  b.x = (float*)ar.take((size_t)B * S * D * 4);
  b.xn = ar.take((size_t)B * S * D * es);
  b.qkv = ar.take((size_t)B * S * 3 * D * es);
  b.ao = ar.take((size_t)B * S * D * es);
  b.h = ar.take((size_t)B * S * c.ffn * es);
  return b;
}
inline int ew(int64_t n) { return (int)std::min<int64_t>((n + 255) / 256, 4096); }
}  // namespace

extern "C" size_t dn_nar_encoder_workspace_bytes(const DnNarEnc* m, int32_t B, int32_t L) {
  if (!m || B <= 0 || L <= 0) return 0;
  Arena a{nullptr, 0, 0};
  (void)plan_enc(m, B, L, a);
  return a.off + 512;
}

extern "C" int dn_nar_encoder_forward(DnNarEnc* m, const float* feats, const int32_t* src_lengths, int32_t B, int32_t L, float* enc_out,
                                      int32_t* out_lengths, void* workspace, size_t workspace_bytes, void* stream) {
  DN_CHECK_ARG(m && feats && src_lengths && enc_out && out_lengths && workspace, "dn_nar_encoder_forward: null argument");
  const DnNarEncConfig& c = m->cfg;
  const int S1 = L <= 0 ? 0 : (L - 1) / 2 + 1, S = dn_nar_encoder_out_frames(L);
  DN_CHECK_ARG(B > 0 && L > 0 && S >= 1 && S + c.pad + 1 <= c.max_pos, "dn_nar_encoder_forward: B=%d L=%d (%d frames after the subsampler; positional table %d)", B, L,
               S, c.max_pos);
  DN_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "dn_nar_encoder_forward: workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  Arena ar{(char*)workspace, 0, workspace_bytes};
  const EncBufs b = plan_enc(m, B, L, ar);
  if (ar.off > workspace_bytes) {
    dn_set_error("dn_nar_encoder_forward: workspace %zu < required %zu", workspace_bytes, ar.off);
    return DN_EWORKSPACE;
  }
  const int dtype = c.dtype, es = esize(dtype), D = c.dim, F = c.ffn, H = c.heads, dh = D / H, Ch = c.conv_channels / 2, k = c.kernel;
  const int K0 = padk(k * c.input_dim), K1 = k * Ch, M = B * S;
  // ---- Conv1dSubsampler: conv(k, stride 2) as a contraction over gathered rows, GLU folded into the next gather / the embedding
  hipLaunchKernelGGL((strided_rows_kernel<false>), dim3(ew((int64_t)B * S1 * K0)), dim3(256), 0, s, feats, B, L, S1, c.input_dim, k, K0, b.cols0, dtype);
  {
    DnGemmParams p = gemm_base(dtype, B * S1, c.conv_channels, K0, S1);
    p.terms[0].A = b.cols0; p.terms[0].lda = K0; p.terms[0].W = m->conv0_W;
    p.bias = m->conv0_b; p.out = b.y0; p.ldo = c.conv_channels; p.out_dtype = DN_F32;
    DN_TRY(dn_conv_gemm(&p, s));
  }
  hipLaunchKernelGGL((strided_rows_kernel<true>), dim3(ew((int64_t)M * K1)), dim3(256), 0, s, b.y0, B, S1, S, Ch, k, K1, b.cols1, dtype);
  {
    DnGemmParams p = gemm_base(dtype, M, 2 * D, K1, S);
    p.terms[0].A = b.cols1; p.terms[0].lda = K1; p.terms[0].W = m->conv1_W;
    p.bias = m->conv1_b; p.out = b.y1; p.ldo = 2 * D; p.out_dtype = DN_F32;
    DN_TRY(dn_conv_gemm(&p, s));
  }
  hipLaunchKernelGGL(enc_embed_kernel, dim3(ew((int64_t)M * D)), dim3(256), 0, s, b.y1, src_lengths, 2, B, S, D, m->pos, c.pad, sqrtf((float)D), b.x, out_lengths);
  // ---- pre-norm encoder layers
  auto ln = [&](const float* g, const float* be, void* y, int odt) {
    hipLaunchKernelGGL(layernorm_kernel, dim3((M + 3) / 4), dim3(256), 0, s, b.x, D, y, D, odt, M, D, g, be);
  };
  auto resadd = [&](const void* A, int K, const void* W, const float* bias) -> int {  // x += A W^T + bias
    DnGemmParams p = gemm_base(dtype, M, D, K, S);
    p.terms[0].A = A; p.terms[0].lda = K; p.terms[0].W = W;
    p.bias = bias; p.epilogue = DN_EPI_RESADD; p.res = b.x; p.ldr = D; p.out = b.x; p.ldo = D; p.out_dtype = DN_F32;
    return dn_conv_gemm(&p, s);
  };
  const int sdt = side_dt(dtype), ses = esize(sdt) == 4 ? 4 : es;
  for (int l = 0; l < c.layers; ++l) {
    const float* g = m->ln_g + (size_t)l * 2 * D;
    const float* be = m->ln_b + (size_t)l * 2 * D;
    ln(g, be, b.xn, dtype);  // self_attn_layer_norm (:194-196)
    {
      DnGemmParams p = gemm_base(dtype, M, 3 * D, D, S);  // q ; k ; v projections with their biases
      p.terms[0].A = b.xn; p.terms[0].lda = D; p.terms[0].W = eoff(m->qkv_W, (size_t)l * padn(3 * D) * D, es);
      p.bias = m->qkv_b + (size_t)l * 3 * D; p.out = b.qkv; p.ldo = 3 * D; p.out_dtype = sdt;
      DN_TRY(dn_conv_gemm(&p, s));
    }
    {
      DnAttnParams a;
      memset(&a, 0, sizeof(a));
      a.q = b.qkv; a.k = eoff(b.qkv, D, ses); a.v = eoff(b.qkv, 2 * D, ses); a.out = b.ao;
      a.ldq = a.ldk = a.ldv = 3 * D; a.ldo = D;
      a.B = B; a.T = S; a.heads = H; a.dim_head = dh; a.dtype = dtype; a.lengths = out_lengths;  // key_padding_mask (:197-204)
      a.scale = 1.0f / sqrtf((float)dh);
      DN_TRY(dn_attention(&a, s));
    }
    DN_TRY(resadd(b.ao, D, eoff(m->so_W, (size_t)l * padn(D) * D, es), m->so_b + (size_t)l * D));
    ln(g + D, be + D, b.xn, dtype);  // final_layer_norm (:210-212)
    {
      DnGemmParams p = gemm_base(dtype, M, F, D, S);  // relu(fc1) (:213)
      p.terms[0].A = b.xn; p.terms[0].lda = D; p.terms[0].W = eoff(m->fc1_W, (size_t)l * padn(F) * D, es);
      p.bias = m->fc1_b + (size_t)l * F; p.epilogue = DN_EPI_RELU; p.out = b.h; p.ldo = F;
      DN_TRY(dn_conv_gemm(&p, s));
    }
    DN_TRY(resadd(b.h, F, eoff(m->fc2_W, (size_t)l * padn(D) * F, es), m->fc2_b + (size_t)l * D));
  }
  ln(m->fin_g, m->fin_b, enc_out, DN_F32);  // encoder.layer_norm (:361-362), fp32 [B, S, D]
  DN_CHECK_LAUNCH("dn_nar_encoder_forward");
  return DN_OK;
}
