// The model inside the mask-predict loop downstream of the normalised units: the DECODER side of the reference's
// NAR S2UT model, NARS2UTTransformerModel (research/TranSpeech/nar_transformer.py:569-976) -- TransformerUnitDecoder.forward
// (:321-420) = fairseq's TransformerDecoder with full context (fairseq/models/transformer/transformer_decoder.py:219-330): scaled
// token embedding + sinusoidal positions, pre-norm layers of self-attention, encoder attention and a ReLU FFN
// (fairseq/modules/transformer_layer.py:389-520), final LayerNorm, the 1004-way output projection; and the length predictor
// (:436-480).  The speech encoder is out of scope: its output is a given tensor.  Everything is a sequence of dn_conv_gemm /
// dn_attention launches plus three small kernels here (embedding, LayerNorm, masked mean); the keys / values of the encoder
// attention depend on the encoder output only, so all layers' are one grouped contraction per utterance batch, reused by every
// refinement iteration (dn_nar_cross_kv).  Nothing allocates or synchronises.
#include <new>

#include "common.h"
#include "engine.h"
#include "nar_common.h"

using namespace dn;

namespace dn {

// x[b, t, :] = sqrt(D) * E[tok] + P[pos], pos = padding_idx + (rank of t among the non-pad tokens of its row), pad -> padding_idx
// (fairseq/utils.py:256-266; table row padding_idx is zero; transformer_decoder.py:254-275).  One workgroup per sequence; also
// writes the number of non-pad tokens (the self-attention's key mask: non-pad tokens are a prefix in this loop, :861-868).
__global__ __launch_bounds__(256) void nar_embed_kernel(const int32_t* __restrict__ tokens, int T, const float* __restrict__ emb,
                                                        const float* __restrict__ pos_table, int D, int Dp, int pad, float scale,
                                                        float* __restrict__ x, int32_t* __restrict__ tlen) {
  __shared__ int s_pos[2048];
  __shared__ int s_cnt;
  const int b = blockIdx.x;
  const int32_t* tk = tokens + (int64_t)b * T;
  if (threadIdx.x == 0) {
    int c = 0;
    for (int t = 0; t < T; ++t) {
      const bool np = tk[t] != pad;
      c += np ? 1 : 0;
      s_pos[t] = np ? c + pad : pad;
    }
    s_cnt = c;
    tlen[b] = c;
  }
  __syncthreads();
  const int q = Dp >> 2;
  for (int i = threadIdx.x; i < T * q; i += 256) {
    const int t = i / q, c = (i - t * q) * 4;
    float4 v = make_float4(0, 0, 0, 0);
    if (c < D) {
      const float4 e = *reinterpret_cast<const float4*>(emb + (int64_t)tk[t] * D + c);
      const float4 p = *reinterpret_cast<const float4*>(pos_table + (int64_t)s_pos[t] * D + c);
      v = make_float4(fmaf(scale, e.x, p.x), fmaf(scale, e.y, p.y), fmaf(scale, e.z, p.z), fmaf(scale, e.w, p.w));
    }
    *reinterpret_cast<float4*>(x + ((int64_t)b * T + t) * Dp + c) = v;
  }
}

// _mean_pooling (fairseq/models/nat/nonautoregressive_transformer.py:20-34): masked mean over the source frames; enc fp32 [B, S, D]
__global__ __launch_bounds__(256) void nar_mean_pool_kernel(const float* __restrict__ enc, const int32_t* __restrict__ slen, int S, int D, int Dp,
                                                            float* __restrict__ out) {
  const int b = blockIdx.x;
  const int n = min(slen[b], S);
  for (int c = threadIdx.x; c < Dp; c += 256) {
    float s = 0.f;
    if (c < D)
      for (int t = 0; t < n; ++t) s += enc[((int64_t)b * S + t) * D + c] / (float)n;  // the reference divides before it sums
    out[(int64_t)b * Dp + c] = s;
  }
}

}  // namespace dn

struct DnNar {
  DnNarConfig cfg;
  // packed tensors in table order
  const float *emb, *pos, *len_W;
  const void *qkv_W, *so_W, *cq_W, *ckv_W, *co_W, *fc1_W, *fc2_W, *out_W;
  const float *qkv_b, *so_b, *cq_b, *ckv_b, *co_b, *fc1_b, *fc2_b, *ln_g, *ln_b, *fin_g, *fin_b;
};
static const int kNarTensors = 22;

extern "C" int dn_nar_create(const DnNarConfig* cfg, const void* const* w, int32_t n, DnNar** out) {
  DN_CHECK_ARG(cfg && w && out, "dn_nar_create: null argument");
  DN_CHECK_ARG(n == kNarTensors, "dn_nar_create: expected %d packed tensors, got %d", kNarTensors, n);
  DN_CHECK_ARG(cfg->dtype == DN_F32 || cfg->dtype == DN_BF16 || cfg->dtype == DN_BF16X3 || cfg->dtype == DN_F16, "dn_nar_create: bad dtype");
  DN_CHECK_ARG(cfg->dim > 0 && cfg->dim % 64 == 0 && cfg->dim <= 1024 && cfg->dim % cfg->heads == 0 && (cfg->dim / cfg->heads) % 4 == 0,
               "dn_nar_create: embed dim %d must be a multiple of 64 (<= 1024) and of the head count", cfg->dim);
  DN_CHECK_ARG(cfg->ffn % 64 == 0 && cfg->vocab % 4 == 0 && cfg->layers >= 1 && cfg->layers <= DN_MAX_TERMS, "dn_nar_create: ffn %% 64, vocab %% 4, 1..%d layers", DN_MAX_TERMS);
  for (int i = 0; i < n; ++i) DN_CHECK_ARG(w[i] != nullptr, "dn_nar_create: packed tensor %d is null", i);
  DnNar* m = new (std::nothrow) DnNar();
  DN_CHECK_ARG(m != nullptr, "dn_nar_create: out of host memory");
  m->cfg = *cfg;
  int i = 0;
  m->emb = (const float*)w[i++]; m->pos = (const float*)w[i++]; m->len_W = (const float*)w[i++];
  m->qkv_W = w[i++]; m->qkv_b = (const float*)w[i++]; m->so_W = w[i++]; m->so_b = (const float*)w[i++];
  m->cq_W = w[i++]; m->cq_b = (const float*)w[i++]; m->ckv_W = w[i++]; m->ckv_b = (const float*)w[i++];
  m->co_W = w[i++]; m->co_b = (const float*)w[i++]; m->fc1_W = w[i++]; m->fc1_b = (const float*)w[i++];
  m->fc2_W = w[i++]; m->fc2_b = (const float*)w[i++]; m->ln_g = (const float*)w[i++]; m->ln_b = (const float*)w[i++];
  m->fin_g = (const float*)w[i++]; m->fin_b = (const float*)w[i++]; m->out_W = w[i++];
  *out = m;
  return DN_OK;
}

extern "C" void dn_nar_destroy(DnNar* m) { delete m; }

namespace {
inline int side_dt(int dtype) { return dtype == DN_BF16X3 ? DN_F32 : dtype; }  // q / k / v are read by the attention kernel, not staged

struct NarBufs { float* x; void *xn, *qkv, *cq, *ao, *h; int32_t* tlen; };
NarBufs plan_nar(const DnNar* m, int M, int B, Arena& ar) {
  const int es = esize(m->cfg.dtype), D = m->cfg.dim, F = m->cfg.ffn;
  NarBufs b;
  b.x = (float*)ar.take((size_t)M * D * 4);
  b.xn = ar.take((size_t)M * D * es);
  b.qkv = ar.take((size_t)M * 3 * D * es);
  b.cq = ar.take((size_t)M * D * es);
  b.ao = ar.take((size_t)M * D * es);
  b.h = ar.take((size_t)M * F * es);
  b.tlen = (int32_t*)ar.take((size_t)B * 4 + 64);
  return b;
}
}  // namespace

extern "C" size_t dn_nar_workspace_bytes(const DnNar* m, int32_t B, int32_t T, int32_t S) {
  if (!m || B <= 0 || T <= 0 || S <= 0) return 0;
  Arena a{nullptr, 0, 0};
  (void)plan_nar(m, B * T, B, a);
  const size_t pre = (size_t)B * S * m->cfg.dim * 4 + (size_t)B * (m->cfg.dim + 256) * 4 + 4096;  // dn_nar_cross_kv / dn_nar_predict_lengths staging
  return (a.off > pre ? a.off : pre) + 512;
}

extern "C" size_t dn_nar_cross_kv_bytes(const DnNar* m, int32_t B, int32_t S) {
  return m ? (size_t)m->cfg.layers * B * S * 2 * m->cfg.dim * esize(m->cfg.dtype) + 256 : 0;
}

extern "C" int dn_nar_cross_kv(DnNar* m, const float* enc_out, int32_t B, int32_t S, void* ckv, void* workspace, size_t workspace_bytes, void* stream) {
  DN_CHECK_ARG(m && enc_out && ckv && workspace && B > 0 && S > 0, "dn_nar_cross_kv: bad argument");
  DN_CHECK_ARG(((uintptr_t)workspace & 255) == 0 && ((uintptr_t)ckv & 127) == 0, "dn_nar_cross_kv: buffers must be 256 / 128-byte aligned");
  const int dtype = m->cfg.dtype, es = esize(dtype), D = m->cfg.dim, L = m->cfg.layers;
  DN_CHECK_ARG(workspace_bytes >= (size_t)B * S * D * es, "dn_nar_cross_kv: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  DN_TRY(dn_convert_rows(enc_out, DN_F32, D, workspace, dtype, D, B * S, D, s));
  DnGemmParams p = gemm_base(dtype, B * S, 2 * D, D, S);  // encoder_attn.{k,v}_proj of every layer (transformer_layer.py:455-470)
  p.groups = L;
  p.terms[0].A = workspace; p.terms[0].lda = D; p.terms[0].a_gstride = 0;
  p.terms[0].W = m->ckv_W; p.terms[0].w_gstride = (int64_t)padn(2 * D) * D;
  p.bias = m->ckv_b; p.bias_gstride = 2 * D;
  p.out = ckv; p.ldo = 2 * D; p.out_gstride = (int64_t)B * S * 2 * D; p.out_dtype = side_dt(dtype);
  return dn_conv_gemm(&p, s);
}

extern "C" int dn_nar_predict_lengths(DnNar* m, const float* enc_out, const int32_t* src_lengths, int32_t B, int32_t S, int32_t* lengths,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  DN_CHECK_ARG(m && enc_out && src_lengths && lengths && workspace && B > 0 && S > 0, "dn_nar_predict_lengths: bad argument");
  const int D = m->cfg.dim;
  DN_CHECK_ARG(workspace_bytes >= (size_t)B * (D + 256) * 4 + 256, "dn_nar_predict_lengths: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* pooled = (float*)workspace;
  float* logit = pooled + (size_t)B * D;
  hipLaunchKernelGGL(nar_mean_pool_kernel, dim3(B), dim3(256), 0, s, enc_out, src_lengths, S, D, D, pooled);
  DnGemmParams p = gemm_base(DN_F32, B, 256, D, 1);  // F.linear(enc_feats, embed_length.weight) (:443), exact fp32; arg-max of the log-softmax = of the logits
  p.terms[0].A = pooled; p.terms[0].lda = D; p.terms[0].W = m->len_W;
  p.out = logit; p.ldo = 256; p.out_dtype = DN_F32;
  DN_TRY(dn_conv_gemm(&p, s));
  return dn_argmax_units(logit, 256, B, 256, 0, lengths, s);
}

extern "C" int dn_nar_decoder_forward(DnNar* m, const int32_t* tokens, const void* ckv, const int32_t* src_lengths, int32_t B, int32_t T, int32_t S,
                                      float* logits, void* workspace, size_t workspace_bytes, void* stream) {
  DN_CHECK_ARG(m && tokens && ckv && src_lengths && logits && workspace, "dn_nar_decoder_forward: null argument");
  DN_CHECK_ARG(B > 0 && T > 0 && T <= 2048 && T + m->cfg.pad + 1 <= m->cfg.max_pos && S > 0, "dn_nar_decoder_forward: B=%d T=%d S=%d (T <= 2048 and within the positional table)", B, T, S);
  DN_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "dn_nar_decoder_forward: workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const DnNarConfig& c = m->cfg;
  const int dtype = c.dtype, es = esize(dtype), M = B * T, D = c.dim, F = c.ffn, L = c.layers, H = c.heads, dh = D / H, V = c.vocab;
  Arena ar{(char*)workspace, 0, workspace_bytes};
  const NarBufs b = plan_nar(m, M, B, ar);
  if (ar.off > workspace_bytes) {
    dn_set_error("dn_nar_decoder_forward: workspace %zu < required %zu", workspace_bytes, ar.off);
    return DN_EWORKSPACE;
  }
  hipLaunchKernelGGL(nar_embed_kernel, dim3(B), dim3(256), 0, s, tokens, T, m->emb, m->pos, D, D, c.pad, sqrtf((float)D), b.x, b.tlen);
  auto ln = [&](const float* g, const float* be) {
    hipLaunchKernelGGL(layernorm_kernel, dim3((M + 3) / 4), dim3(256), 0, s, b.x, D, b.xn, D, dtype, M, D, g, be);
  };
  auto attn = [&](const void* q, int ldq, const void* k, const void* v, int ldkv, int Tk, const int32_t* lens) -> int {
    DnAttnParams a;
    memset(&a, 0, sizeof(a));
    a.q = q; a.k = k; a.v = v; a.out = b.ao;
    a.ldq = ldq; a.ldk = a.ldv = ldkv; a.ldo = D;
    a.B = B; a.T = T; a.Tk = Tk; a.heads = H; a.dim_head = dh; a.dtype = dtype; a.lengths = lens;
    a.scale = 1.0f / sqrtf((float)dh);  // q * head_dim^-0.5 (multihead_attention.py)
    return dn_attention(&a, s);
  };
  auto resadd = [&](const void* A, int K, const void* W, const float* bias) -> int {  // x += A W^T + bias
    DnGemmParams p = gemm_base(dtype, M, D, K, T);
    p.terms[0].A = A; p.terms[0].lda = K; p.terms[0].W = W;
    p.bias = bias; p.epilogue = DN_EPI_RESADD; p.res = b.x; p.ldr = D; p.out = b.x; p.ldo = D; p.out_dtype = DN_F32;
    return dn_conv_gemm(&p, s);
  };
  const int sdt = side_dt(dtype), ses = esize(sdt) == 4 ? 4 : es;
  for (int l = 0; l < L; ++l) {
    const float* g = m->ln_g + (size_t)l * 3 * D;
    const float* be = m->ln_b + (size_t)l * 3 * D;
    ln(g, be);  // self_attn_layer_norm (pre-norm, :421-423)
    {
      DnGemmParams p = gemm_base(dtype, M, 3 * D, D, T);  // q ; k ; v projections with their biases
      p.terms[0].A = b.xn; p.terms[0].lda = D; p.terms[0].W = eoff(m->qkv_W, (size_t)l * padn(3 * D) * D, es);
      p.bias = m->qkv_b + (size_t)l * 3 * D; p.out = b.qkv; p.ldo = 3 * D; p.out_dtype = sdt;
      DN_TRY(dn_conv_gemm(&p, s));
    }
    DN_TRY(attn(b.qkv, 3 * D, eoff(b.qkv, D, ses), eoff(b.qkv, 2 * D, ses), 3 * D, 0, b.tlen));  // full context, keys = non-pad tokens
    DN_TRY(resadd(b.ao, D, eoff(m->so_W, (size_t)l * padn(D) * D, es), m->so_b + (size_t)l * D));
    ln(g + D, be + D);  // encoder_attn_layer_norm (:447-449)
    {
      DnGemmParams p = gemm_base(dtype, M, D, D, T);
      p.terms[0].A = b.xn; p.terms[0].lda = D; p.terms[0].W = eoff(m->cq_W, (size_t)l * padn(D) * D, es);
      p.bias = m->cq_b + (size_t)l * D; p.out = b.cq; p.ldo = D; p.out_dtype = sdt;
      DN_TRY(dn_conv_gemm(&p, s));
    }
    const void* kv = eoff(ckv, (size_t)l * B * S * 2 * D, ses);
    DN_TRY(attn(b.cq, D, kv, eoff(kv, D, ses), 2 * D, S, src_lengths));
    DN_TRY(resadd(b.ao, D, eoff(m->co_W, (size_t)l * padn(D) * D, es), m->co_b + (size_t)l * D));
    ln(g + 2 * D, be + 2 * D);  // final_layer_norm (:495-497)
    {
      DnGemmParams p = gemm_base(dtype, M, F, D, T);  // relu(fc1) (:499-501)
      p.terms[0].A = b.xn; p.terms[0].lda = D; p.terms[0].W = eoff(m->fc1_W, (size_t)l * padn(F) * D, es);
      p.bias = m->fc1_b + (size_t)l * F; p.epilogue = DN_EPI_RELU; p.out = b.h; p.ldo = F;
      DN_TRY(dn_conv_gemm(&p, s));
    }
    DN_TRY(resadd(b.h, F, eoff(m->fc2_W, (size_t)l * padn(D) * F, es), m->fc2_b + (size_t)l * D));
  }
  ln(m->fin_g, m->fin_b);  // decoder.layer_norm (transformer_decoder.py:316-317)
  DnGemmParams p = gemm_base(dtype, M, V, D, T);  // output_projection (no bias)
  p.terms[0].A = b.xn; p.terms[0].lda = D; p.terms[0].W = m->out_W;
  p.out = logits; p.ldo = V; p.out_dtype = DN_F32;
  DN_TRY(dn_conv_gemm(&p, s));
  DN_CHECK_LAUNCH("dn_nar_decoder_forward");
  return DN_OK;
}
