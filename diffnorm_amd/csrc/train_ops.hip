// Training-side HBM-bound kernels: the backward of every non-contraction op on the path, the loss
// gradients, and the reductions that turn per-frame quantities into parameter gradients.
//   gate        tanh(h) * sigmoid(h) + res and its derivative            reference latent_module.py:525-530
//   GEGLU       gelu(gate) * value on the packed [8 value ; 8 gate] columns                 :881-884
//   RMSNorm     x / max(|x|, 1e-12) * sqrt(D) * gamma [* g_c + b_c]                          :620-639
//   posterior   z = mean + exp(.5 logvar) * noise, KL                      distributions.py:24-41, 62-74
//   LS-CE       label-smoothed NLL over log_softmax(logits)   fairseq/criterions/label_smoothed_cross_entropy.py:34-51
//   masked MSE  mean over the valid elements                                latent_module.py:1135-1138
// Reductions over frames (bias / gamma gradients, loss sums) are two-stage with fixed block counts: no atomics, one
// summation order per shape, so a training step is bit-reproducible.
#include "common.h"

#include <algorithm>

namespace dn {

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline int ew_blocks(int64_t n4) {
  const int64_t b = (n4 + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// d/dh [tanh(h) sigmoid(h)] = sech^2(h) sigmoid(h) + tanh(h) sigmoid(h) (1 - sigmoid(h)), from one exponential
__device__ __forceinline__ float gate_grad(float h) {
  const float hc = fminf(fmaxf(h, -30.0f), 30.0f);
  const float q = __expf(-hc);
  const float q2 = q * q;
  const float sg = fast_rcp(1.0f + q);
  const float th = (1.0f - q2) * fast_rcp(1.0f + q2);
  return (1.0f - th * th) * sg + th * sg * (1.0f - sg);
}

// exact-GELU derivative Phi(x) + x phi(x), Phi from the same A-S 7.1.26 erf the forward epilogue uses
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = fast_rcp(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = __expf(-z * z);
  const float erf_abs = 1.0f - poly * e;
  const float Phi = 0.5f * (1.0f + copysignf(erf_abs, x));
  return Phi + x * 0.39894228040143267794f * e;  // exp(-x^2/2) = exp(-z^2)
}

// ------------------------------------------------------------------------------------------ gate
// out = tanh(h') sigmoid(h') + res,  h' = h * gamma[b] + beta[b] when gb != NULL (FiLM, :525-527) else h.
// All tensors [M, ld] with the same ld (padded channels are zeros and stay zeros).
__global__ __launch_bounds__(256) void gate_fwd_kernel(const void* __restrict__ h, const void* __restrict__ res, void* __restrict__ out,
                                                       int dtype, int M, int ld, int T, const float* __restrict__ gb, int gb_ld,
                                                       int gb_half) {
  const int64_t n4 = (int64_t)M * ld / 4;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int64_t e = i * 4;
    float4 hv = load4(h, e, dtype);
    const float4 rv = load4(res, e, dtype);
    if (gb) {
      const int m = (int)(e / ld), c = (int)(e - (int64_t)m * ld);
      const float* g = gb + (int64_t)(m / T) * gb_ld + c;
      const float4 ga = *reinterpret_cast<const float4*>(g), be = *reinterpret_cast<const float4*>(g + gb_half);
      hv = make_float4(hv.x * ga.x + be.x, hv.y * ga.y + be.y, hv.z * ga.z + be.z, hv.w * ga.w + be.w);
    }
    store4(out, e, dtype, tanh_sigmoid_gate(hv.x) + rv.x, tanh_sigmoid_gate(hv.y) + rv.y, tanh_sigmoid_gate(hv.z) + rv.z,
           tanh_sigmoid_gate(hv.w) + rv.w);
  }
}

// dh = dout * gate'(h') [* gamma]; with FiLM also the per-frame products the caller reduces over t into the conditioning
// gradients: dgb_rows[m, c] = dg * h (-> d gamma) and dgb_rows[m, gb_half + c] = dg (-> d beta), dg = dout * gate'(h').
__global__ __launch_bounds__(256) void gate_bwd_kernel(const void* __restrict__ dout, const void* __restrict__ h, void* __restrict__ dh,
                                                       int dtype, int M, int ld, int T, const float* __restrict__ gb, int gb_ld,
                                                       int gb_half, float* __restrict__ dgb_rows, int dgb_ld) {
  const int64_t n4 = (int64_t)M * ld / 4;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int64_t e = i * 4;
    const float4 hv = load4(h, e, dtype);
    const float4 dv = load4(dout, e, dtype);
    if (gb) {
      const int m = (int)(e / ld), c = (int)(e - (int64_t)m * ld);
      const float* g = gb + (int64_t)(m / T) * gb_ld + c;
      const float4 ga = *reinterpret_cast<const float4*>(g), be = *reinterpret_cast<const float4*>(g + gb_half);
      const float4 dg = make_float4(dv.x * gate_grad(hv.x * ga.x + be.x), dv.y * gate_grad(hv.y * ga.y + be.y),
                                    dv.z * gate_grad(hv.z * ga.z + be.z), dv.w * gate_grad(hv.w * ga.w + be.w));
      store4(dh, e, dtype, dg.x * ga.x, dg.y * ga.y, dg.z * ga.z, dg.w * ga.w);
      if (dgb_rows) {
        float* r = dgb_rows + (int64_t)m * dgb_ld + c;
        *reinterpret_cast<float4*>(r) = make_float4(dg.x * hv.x, dg.y * hv.y, dg.z * hv.z, dg.w * hv.w);
        *reinterpret_cast<float4*>(r + gb_half) = dg;
      }
    } else {
      store4(dh, e, dtype, dv.x * gate_grad(hv.x), dv.y * gate_grad(hv.y), dv.z * gate_grad(hv.z), dv.w * gate_grad(hv.w));
    }
  }
}

// ------------------------------------------------------------------------------------------ GEGLU
// pre [M, 2*ip] in the packed column order of the GEGLU projection (per 16 columns: 8 value, then the 8 gates of the same
// outputs); out [M, ip].  ip % 8 == 0.
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const void* __restrict__ pre, void* __restrict__ out, int dtype, int M, int ip) {
  const int64_t n4 = (int64_t)M * ip / 4;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int64_t e = i * 4;
    const int m = (int)(e / ip), j = (int)(e - (int64_t)m * ip);
    const int64_t pv = (int64_t)m * 2 * ip + (j >> 3) * 16 + (j & 7);
    const float4 v = load4(pre, pv, dtype), g = load4(pre, pv + 8, dtype);
    store4(out, e, dtype, gelu_erf(g.x) * v.x, gelu_erf(g.y) * v.y, gelu_erf(g.z) * v.z, gelu_erf(g.w) * v.w);
  }
}

__global__ __launch_bounds__(256) void geglu_bwd_kernel(const void* __restrict__ dout, const void* __restrict__ pre, void* __restrict__ dpre,
                                                        int dtype, int M, int ip) {
  const int64_t n4 = (int64_t)M * ip / 4;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int64_t e = i * 4;
    const int m = (int)(e / ip), j = (int)(e - (int64_t)m * ip);
    const int64_t pv = (int64_t)m * 2 * ip + (j >> 3) * 16 + (j & 7);
    const float4 v = load4(pre, pv, dtype), g = load4(pre, pv + 8, dtype), d = load4(dout, e, dtype);
    store4(dpre, pv, dtype, d.x * gelu_erf(g.x), d.y * gelu_erf(g.y), d.z * gelu_erf(g.z), d.w * gelu_erf(g.w));
    store4(dpre, pv + 8, dtype, d.x * v.x * gelu_erf_grad(g.x), d.y * v.y * gelu_erf_grad(g.y), d.z * v.z * gelu_erf_grad(g.z),
           d.w * v.w * gelu_erf_grad(g.w));
  }
}

// ------------------------------------------------------------------------------------------ RMSNorm backward
// y = x * r * s * G + Bc with r = 1 / max(|x|, 1e-12), s = sqrt(D), G = gamma (learned) or g_c[b] (adaptive).
//   dx = s r (G dy - x r^2 sum_c(G dy x))  [+ dres]
//   learned:  dgamma[c]   = sum_rows dy x r s
//   adaptive: dg_c[b][c]  = sum_t dy x r s ,  db_c[b][c] = sum_t dy
// One wave per row, D <= 1024 (the row lives in registers).  A workgroup walks `rows_per_block` consecutive rows of ONE
// sample and keeps the gamma / (gamma, beta) partial sums of its columns in registers; partial[blk][...] is reduced by
// colsum_final_kernel in a fixed order.
__global__ __launch_bounds__(256, 3) void rmsnorm_bwd_kernel(const float* __restrict__ x, int ldx, const void* __restrict__ dy, int lddy,
                                                          int dy_dtype, int B, int T, int D, const float* __restrict__ gamma,
                                                          const float* __restrict__ gb, int gb_ld, int gb_half,
                                                          const float* __restrict__ dres, float* __restrict__ dx, void* __restrict__ dx_act,
                                                          int act_dtype, int ld_act, int rows_per_block, int blocks_per_sample,
                                                          float* __restrict__ partial, int part_ld) {
  __shared__ float red[4][2 * 1024];
  const int b = blockIdx.x / blocks_per_sample, chunk = blockIdx.x - b * blocks_per_sample;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t0 = chunk * rows_per_block, t1 = min(T, t0 + rows_per_block);
  const float s = sqrtf((float)D);
  const float* gbr = gb ? gb + (int64_t)b * gb_ld : nullptr;
  float4 G[4];
  float4 accg[4], accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (i * 64 + lane) * 4;
    G[i] = make_float4(1.f, 1.f, 1.f, 1.f);
    if (c < D) {
      if (gamma) G[i] = *reinterpret_cast<const float4*>(gamma + c);
      if (gbr) {
        const float4 gc = *reinterpret_cast<const float4*>(gbr + c);
        G[i] = make_float4(G[i].x * gc.x, G[i].y * gc.y, G[i].z * gc.z, G[i].w * gc.w);
      }
    }
    accg[i] = accb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // (the residual gradient of a row is requested with the row itself, not after the row's reductions: one memory round trip per row
  // instead of two.  Keeping a second row in flight per wave cost 100 registers and a workgroup per CU: 47 -> 58 us.)
  for (int t = t0 + wave; t < t1; t += 4) {
    const int64_t row = (int64_t)b * T + t;
    float4 xv[4], dv[4], rv[4];
    float ss = 0.f, dot = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = (i * 64 + lane) * 4;
      xv[i] = dv[i] = rv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < D) {
        xv[i] = *reinterpret_cast<const float4*>(x + row * ldx + c);
        dv[i] = load4(dy, row * lddy + c, dy_dtype);
        if (dres) rv[i] = *reinterpret_cast<const float4*>(dres + row * ldx + c);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ss += xv[i].x * xv[i].x + xv[i].y * xv[i].y + xv[i].z * xv[i].z + xv[i].w * xv[i].w;
      dot += G[i].x * dv[i].x * xv[i].x + G[i].y * dv[i].y * xv[i].y + G[i].z * dv[i].z * xv[i].z + G[i].w * dv[i].w * xv[i].w;
    }
    ss = wave_sum64(ss);
    dot = wave_sum64(dot);
    const float r = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    const float sr = s * r, k = r * r * dot;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c >= D) continue;
      float4 o = make_float4(sr * (G[i].x * dv[i].x - xv[i].x * k), sr * (G[i].y * dv[i].y - xv[i].y * k),
                             sr * (G[i].z * dv[i].z - xv[i].z * k), sr * (G[i].w * dv[i].w - xv[i].w * k));
      o = make_float4(o.x + rv[i].x, o.y + rv[i].y, o.z + rv[i].z, o.w + rv[i].w);
      *reinterpret_cast<float4*>(dx + row * ldx + c) = o;
      if (dx_act) store4(dx_act, row * ld_act + c, act_dtype, o.x, o.y, o.z, o.w);
      accg[i].x += dv[i].x * xv[i].x * sr; accg[i].y += dv[i].y * xv[i].y * sr;
      accg[i].z += dv[i].z * xv[i].z * sr; accg[i].w += dv[i].w * xv[i].w * sr;
      accb[i].x += dv[i].x; accb[i].y += dv[i].y; accb[i].z += dv[i].z; accb[i].w += dv[i].w;
    }
    if (dx_act)  // pad columns of the operand copy
      for (int c = D + lane * 4; c < ld_act; c += 256) store4(dx_act, row * ld_act + c, act_dtype, 0.f, 0.f, 0.f, 0.f);
    for (int c = D + lane * 4; c < ldx; c += 256) *reinterpret_cast<float4*>(dx + row * ldx + c) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (!partial) return;
  // the four waves' column sums -> one partial row per workgroup: [dgamma (D) | dbeta (D, adaptive only)]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < D) {
      *reinterpret_cast<float4*>(&red[wave][c]) = accg[i];
      *reinterpret_cast<float4*>(&red[wave][1024 + c]) = accb[i];
    }
  }
  __syncthreads();
  float* pr = partial + (int64_t)blockIdx.x * part_ld;
  for (int c = threadIdx.x; c < D; c += 256) {
    float g = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    if (gbr) {
      // adaptive: d g_c = sum dy * (x r s [* gamma]) ; the learned gamma is absent for conditional norms (:662-663)
      pr[c] = g;
      pr[gb_half + c] = (red[0][1024 + c] + red[1][1024 + c]) + (red[2][1024 + c] + red[3][1024 + c]);
    } else {
      pr[c] = g;
    }
  }
}

// out[g, c] (+)= scale * sum_{j < per_group} partial[(g * per_group + j), c].  A workgroup (1024 threads) owns 64 columns of one
// group; its sixteen waves take the rows j = w, w + 16, ... (coalesced 256-byte reads) and their sums are added in a fixed order.
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ partial, int part_ld, int per_group, int C,
                                                            float* __restrict__ out, int out_ld, float scale, int accumulate) {
  __shared__ float red[16][64];
  const int g = blockIdx.y;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    const float* p = partial + (int64_t)g * per_group * part_ld + c;
    int j = ty;
    for (; j + 16 < per_group; j += 32) {
      s0 += p[(int64_t)j * part_ld];
      s1 += p[(int64_t)(j + 16) * part_ld];
    }
    if (j < per_group) s0 += p[(int64_t)j * part_ld];
  }
  red[ty][tx] = s0 + s1;
  __syncthreads();
  if (ty == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w][tx];
    float* o = out + (int64_t)g * out_ld + c;
    *o = accumulate ? *o + t * scale : t * scale;
  }
}

// partial[blk, c] = sum over the block's row range of src[m, c]; rows [g * rows_per_group, (g+1) * rows_per_group) form
// group g, each split into `chunks` blocks (blockIdx.y = g * chunks + chunk).  A workgroup owns 256 columns: 32 threads of 8 columns
// (a wave row reads 512 contiguous bytes of a 2-byte tensor) x 8 row lanes that stride the block's rows with four loads in flight,
// summed across the row lanes through LDS in a fixed order.  (The first form -- one thread per 4 columns walking its 48 rows one
// 8-byte load at a time on 256 workgroups -- ran at 0.9-2.2 TB/s: 6.5 % of a VAE training update went to bias gradients.)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const void* __restrict__ src, int ld, int dtype, int rows_per_group, int chunks,
                                                             int C, float* __restrict__ partial, int part_ld) {
  __shared__ float red[8][256];
  const int g = blockIdx.y / chunks, chunk = blockIdx.y - g * chunks;
  const int per = (rows_per_group + chunks - 1) / chunks;
  const int r0 = chunk * per, r1 = min(rows_per_group, r0 + per);
  const int ct = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 256 + ct * 8;
  const bool lo = c < C, hi = c + 4 < C;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  if (lo) {
    const int64_t base = (int64_t)g * rows_per_group * ld + c;
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += 8) {
      const float4 v = load4(src, base + (int64_t)r * ld, dtype);
      s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
      if (hi) {
        const float4 w = load4(src, base + (int64_t)r * ld + 4, dtype);
        s1.x += w.x; s1.y += w.y; s1.z += w.z; s1.w += w.w;
      }
    }
  }
  *reinterpret_cast<float4*>(&red[rl][ct * 8]) = s0;
  *reinterpret_cast<float4*>(&red[rl][ct * 8 + 4]) = s1;
  __syncthreads();
  const int cc = blockIdx.x * 256 + threadIdx.x;
  if (cc < C) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) t += red[w][threadIdx.x];
    partial[(int64_t)blockIdx.y * part_ld + cc] = t;
  }
}

// ------------------------------------------------------------------------------------------ posterior backward
// dparams[m, :] = [d mean ; d logvar] in act dtype [M, ldo] (pad columns zero):
//   d mean   = dz + klw * mean                      (valid frames only for the KL part)
//   d logvar = (dz * noise * 0.5 * std + klw * 0.5 * (var - 1)) * 1[-30 <= logvar <= 20]
// klw = (criterion weight of the KL) / (B * Z * T): kl_3d's mean over (z, T) per sample, then the batch mean.
__global__ __launch_bounds__(256) void posterior_bwd_kernel(const float* __restrict__ params, int ldp, const float* __restrict__ noise, int ldn,
                                                            const float* __restrict__ dz, int lddz, void* __restrict__ dparams,
                                                            int act_dtype, int ldo, int M, int Z, int T,
                                                            const int32_t* __restrict__ lengths, float klw) {
  const int64_t n = (int64_t)M * ldo;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int m = (int)(i / ldo), c = (int)(i - (int64_t)m * ldo);
    float v = 0.f;
    if (c < 2 * Z) {
      const int b = m / T, t = m - b * T;
      const float kw = (lengths ? t < lengths[b] : true) ? klw : 0.f;
      const int cz = c < Z ? c : c - Z;
      const float g = dz[(int64_t)m * lddz + cz];
      if (c < Z) {
        v = g + kw * params[(int64_t)m * ldp + c];
      } else {
        const float lv = params[(int64_t)m * ldp + c];
        if (lv >= -30.0f && lv <= 20.0f) {
          const float sd = expf(0.5f * lv);
          v = g * noise[(int64_t)m * ldn + cz] * 0.5f * sd + kw * 0.5f * (sd * sd - 1.0f);
        }
      }
    }
    if (act_dtype == DN_BF16)
      reinterpret_cast<uint16_t*>(dparams)[i] = (uint16_t)(pack_bf16x2(v, 0.f) & 0xffff);
    else
      reinterpret_cast<float*>(dparams)[i] = v;
  }
}

// ------------------------------------------------------------------------------------------ label-smoothed CE
// One wave per frame; V <= 1024 (16 logits per lane in registers).  target 0 = pad (ignored).
//   rows[m] = {nll, smooth, correct, valid}: nll = lse - logit[target], smooth = V lse - sum logits (the two sums of
//   label_smoothed_nll_loss), correct = argmax == target.
//   dlogits = gscale * ((1 - eps - eps_i) (p - onehot) + eps_i (V p - 1)), eps_i = eps / (V - 1), in act dtype [M, ldd].
__global__ __launch_bounds__(256) void lsce_kernel(const float* __restrict__ logits, int ld, const int32_t* __restrict__ target, int M, int V,
                                                   float eps, float gscale, float* __restrict__ rows, void* __restrict__ dlogits,
                                                   int act_dtype, int ldd) {
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (m >= M) return;
  const float* lr = logits + (int64_t)m * ld;
  float v[16];
  float mx = -INFINITY, sum = 0.f;
  int best = 0x7fffffff;
  float bestv = -INFINITY;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = i * 64 + lane;
    v[i] = c < V ? lr[c] : -INFINITY;
    if (c < V) {
      sum += v[i];
      if (v[i] > bestv) { bestv = v[i]; best = c; }
    }
    mx = fmaxf(mx, v[i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bestv, o, 64);
    const int oi = __shfl_xor(best, o, 64);
    if (ov > bestv || (ov == bestv && oi < best)) { bestv = ov; best = oi; }
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  }
  sum = wave_sum64(sum);
  float se = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) se += (i * 64 + lane) < V ? expf(v[i] - mx) : 0.f;
  se = wave_sum64(se);
  const float lse = mx + logf(se);
  const int tg = target[m];
  const bool valid = tg != 0;
  if (lane == 0) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) r = make_float4(lse - lr[tg], (float)V * lse - sum, best == tg ? 1.f : 0.f, 1.f);
    *reinterpret_cast<float4*>(rows + (int64_t)m * 4) = r;
  }
  if (!dlogits) return;
  const float eps_i = eps / (float)(V - 1);
  const float a = valid ? gscale * (1.0f - eps - eps_i) : 0.f, bq = valid ? gscale * eps_i : 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = i * 64 + lane;
    if (c >= ldd) continue;
    float g = 0.f;
    if (c < V) {
      const float p = expf(v[i] - lse);
      g = a * (p - (c == tg ? 1.f : 0.f)) + bq * ((float)V * p - 1.f);
    }
    if (act_dtype == DN_BF16)
      reinterpret_cast<uint16_t*>(dlogits)[(int64_t)m * ldd + c] = (uint16_t)(pack_bf16x2(g, 0.f) & 0xffff);
    else
      reinterpret_cast<float*>(dlogits)[(int64_t)m * ldd + c] = g;
  }
}

// ------------------------------------------------------------------------------------------ masked MSE
// Valid frames (t < lengths[b]): sq_rows[m] = sum_c (pred - target)^2 ; dpred[m, c] (+)= gscale * (pred - target); pads: 0.
// gscale = (criterion weight) * 2 / (n_valid_frames * C).  dpred fp32 [M, ldd]; dpred_act its operand copy [M, ld_act].
__global__ __launch_bounds__(256) void masked_mse_kernel(const float* __restrict__ pred, int ldp, const float* __restrict__ target, int ldt,
                                                         int M, int C, int T, const int32_t* __restrict__ lengths, float gscale,
                                                         float* __restrict__ sq_rows, float* __restrict__ dpred, int ldd, int accumulate,
                                                         void* __restrict__ dpred_act, int act_dtype, int ld_act) {
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (m >= M) return;
  const int b = m / T, t = m - b * T;
  const bool valid = lengths ? t < lengths[b] : true;
  float sq = 0.f;
  const int wide = max(C, max(dpred ? ldd : 0, dpred_act ? ld_act : 0));
  for (int c = lane * 4; c < wide; c += 256) {
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C && valid) {
      const float4 p = *reinterpret_cast<const float4*>(pred + (int64_t)m * ldp + c);
      const float4 q = *reinterpret_cast<const float4*>(target + (int64_t)m * ldt + c);
      const float4 d = make_float4(p.x - q.x, p.y - q.y, p.z - q.z, p.w - q.w);
      sq += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
      g = make_float4(gscale * d.x, gscale * d.y, gscale * d.z, gscale * d.w);
    }
    if (dpred && c < ldd) {
      float4* o = reinterpret_cast<float4*>(dpred + (int64_t)m * ldd + c);
      if (accumulate && c < C) {
        const float4 old = *o;
        g = make_float4(g.x + old.x, g.y + old.y, g.z + old.z, g.w + old.w);
      }
      *o = g;
    }
    if (dpred_act && c < ld_act) store4(dpred_act, (int64_t)m * ld_act + c, act_dtype, g.x, g.y, g.z, g.w);
  }
  sq = wave_sum64(sq);
  if (lane == 0 && sq_rows) sq_rows[m] = sq;
}

// ------------------------------------------------------------------------------------------ small helpers
constexpr int kVecSumBlocks = 256;
__global__ __launch_bounds__(256) void vec_sum_partial_kernel(const float* __restrict__ v, int64_t n, float* __restrict__ partial) {
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += v[i];
  s = wave_sum64(s);
  __shared__ float ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}
__global__ __launch_bounds__(256) void vec_sum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int accumulate) {
  float s = partial[threadIdx.x];  // kVecSumBlocks == blockDim
  s = wave_sum64(s);
  __shared__ float ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = (ws[0] + ws[1]) + (ws[2] + ws[3]);
    out[0] = accumulate ? out[0] + t : t;
  }
}

// dst[g][r][c] = sum_{k < count} src[k][g][r][c]-style group sum: dst[i] = sum_k src[k * stride + i]
__global__ __launch_bounds__(256) void sum_groups_kernel(const void* __restrict__ src, int64_t stride, int count, void* __restrict__ dst,
                                                         int dtype, int64_t n) {
  const int64_t n4 = n / 4;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 s = load4(src, i * 4, dtype);
    for (int k = 1; k < count; ++k) {
      const float4 v = load4(src, k * stride + i * 4, dtype);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    store4(dst, i * 4, dtype, s.x, s.y, s.z, s.w);
  }
}

// part[s][b][c] = sum over the slice's rows n of X[b][n] * W[n][c]: a handful of rows (B <= RB) against a huge ROW-MAJOR fp32 matrix
// W [N][C] -- the data gradient of the conditioning projection (d cond = d gb [B, 57 k] . W_c [57 k, 2048], 470 MB of weights).  The
// matrix is streamed exactly once, as it lies (whole rows, float4 per thread, eight rows in flight per thread): no transposed copy
// to make per update and no MFMA tile that is 1/8 full.  A workgroup owns 1024 columns x one slice of rows; the slice's X values go
// through LDS in chunks of 64 rows ([n][b] order: a thread reads its RB multipliers of a row as broadcast float4s).
template <int RB>
__global__ __launch_bounds__(256) void rows_times_weight_kernel(const float* __restrict__ X, int ldx, int B, const float* __restrict__ W, int ldw,
                                                                int N, int C, int per, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float xs[64][RB];
  const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int n_begin = blockIdx.y * per;
  int n_end = n_begin + per;
  n_end = n_end < N ? n_end : N;
  float4 acc[RB];
#pragma unroll
  for (int b = 0; b < RB; ++b) acc[b] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int n0 = n_begin; n0 < n_end; n0 += 64) {
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * RB; i += 256) {  // xs[j][b] = X[b][n0 + j]: consecutive threads read consecutive n of one row b
      const int b = i >> 6, j = i & 63;
      xs[j][b] = (b < B && n0 + j < n_end) ? X[(int64_t)b * ldx + n0 + j] : 0.f;
    }
    __syncthreads();
    if (c < C) {
      const int rows = n_end - n0 < 64 ? n_end - n0 : 64;
      const float* wp = W + (int64_t)n0 * ldw + c;
      int j = 0;
      for (; j + 8 <= rows; j += 8) {
        float4 w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = *reinterpret_cast<const float4*>(wp + (int64_t)(j + u) * ldw);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int b4 = 0; b4 < RB / 4; ++b4) {
            const float4 x = *reinterpret_cast<const float4*>(&xs[j + u][b4 * 4]);
            const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float4& a = acc[b4 * 4 + q];
              a.x = fmaf(xv[q], w[u].x, a.x); a.y = fmaf(xv[q], w[u].y, a.y); a.z = fmaf(xv[q], w[u].z, a.z); a.w = fmaf(xv[q], w[u].w, a.w);
            }
          }
      }
      for (; j < rows; ++j) {
        const float4 w = *reinterpret_cast<const float4*>(wp + (int64_t)j * ldw);
#pragma unroll
        for (int b = 0; b < RB; ++b) {
          const float xv = xs[j][b];
          acc[b].x = fmaf(xv, w.x, acc[b].x); acc[b].y = fmaf(xv, w.y, acc[b].y); acc[b].z = fmaf(xv, w.z, acc[b].z); acc[b].w = fmaf(xv, w.w, acc[b].w);
        }
      }
    }
  }
  if (c < C)
#pragma unroll
    for (int b = 0; b < RB; ++b)
      if (b < B) *reinterpret_cast<float4*>(part + ((int64_t)blockIdx.y * B + b) * C + c) = acc[b];
}

// Batched padded transpose of packed weight matrices (the operand of the data-gradient contraction):
// dst[n][c][r] = src[n][r][c] for r < R, c < Cc; dst matrices are [Cp][Rp] with zeros elsewhere; consecutive matrices lie
// src_stride / dst_stride elements apart (src rows beyond R -- the zero pad rows of a packed weight -- are not read).
template <typename T>
__global__ __launch_bounds__(256) void transpose_mat_kernel(const T* __restrict__ src, int64_t src_stride, int R, int Cc,
                                                            T* __restrict__ dst, int64_t dst_stride, int Rp, int Cp) {
  __shared__ T tile[64][65];
  const T* s = src + (int64_t)blockIdx.z * src_stride;
  T* d = dst + (int64_t)blockIdx.z * dst_stride;
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < Cc) ? s[(int64_t)r * Cc + c] : T(0);
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < Cp && r < Rp) d[(int64_t)c * Rp + r] = tile[tx][i];
  }
}

// Weight-gradient reduction: grad[tap][n][k] += sum_s part[s][n][tap * rows_w + k]  (n < cout, k < Kp)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int slices, int cout, int n_total, int rows_w,
                                                           int n_taps, float* __restrict__ grad, int Np, int Kp) {
  const int64_t total = (int64_t)n_taps * cout * (Kp / 4);
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k4 = (int)(i % (Kp / 4));
    const int n = (int)((i / (Kp / 4)) % cout);
    const int tap = (int)(i / ((int64_t)(Kp / 4) * cout));
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int sl = 0; sl < slices; ++sl) {
      const float4 v = *reinterpret_cast<const float4*>(part + ((int64_t)sl * cout + n) * n_total + tap * rows_w + k4 * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float4* g = reinterpret_cast<float4*>(grad + ((int64_t)tap * Np + n) * Kp + k4 * 4);
    const float4 old = *g;
    *g = make_float4(old.x + s.x, old.y + s.y, old.z + s.z, old.w + s.w);
  }
}

// bias vector broadcast-accumulate: dst[j * ld + c] += src[c] for j < count (the skip convs' biases share one gradient)
__global__ __launch_bounds__(256) void add_broadcast_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int ld, int count) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  for (int j = 0; j < count; ++j) dst[(int64_t)j * ld + c] += src[c];
}

// f32 twin of optim.hip's transpose_pad_kernel (the weight-gradient operands of the exact-fp32 mode)
__global__ __launch_bounds__(256) void transpose_pad_f32_kernel(const float* __restrict__ src, int ld, int B, int T, int C, int front, int Tp,
                                                                float* __restrict__ dst, int rows, int rows_total, int row0, int chunk) {
  __shared__ float tile[64][65];
  const int64_t cols = (int64_t)B * Tp;
  const int64_t j0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int64_t j = j0 + r;
    float v = 0.f;
    if (j < cols) {
      const int b = (int)(j / Tp), t = (int)(j - (int64_t)b * Tp) - front;
      const int c = c0 + tx;
      if (t >= 0 && t < T && c < C) v = src[((int64_t)b * T + t) * ld + c];
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r;
    const int64_t j = j0 + tx;
    if (c < rows && j < cols) dst[((j / chunk) * rows_total + row0 + c) * chunk + j % chunk] = tile[tx][r];
  }
}

// ------------------------------------------------------------------------------------------ time conditioning backward
// cond[b, c] = silu(s[b, c]), s = W f[b] + bias, f[b] = [t, sin(2 pi w t), cos(2 pi w t)] (latent_module.py:104-116, 741-745).
// Stage 1 (one workgroup per (16 outputs, sample)): recompute f and s, ds = dcond * silu'(s) -> ds[B, C].
__global__ __launch_bounds__(256) void time_cond_bwd_ds_kernel(const int32_t* __restrict__ times, const float* __restrict__ wf, int half,
                                                               const float* __restrict__ W, const float* __restrict__ bias, int C,
                                                               const float* __restrict__ dcond, int ldd, float* __restrict__ ds) {
  extern __shared__ float feat[];
  const int b = blockIdx.y;
  const int nfeat = 2 * half + 1;
  const float tf = (float)times[b];
  for (int k = threadIdx.x; k < nfeat; k += 256) {
    float f = tf;
    if (k > 0) {
      const int j = k <= half ? k - 1 : k - 1 - half;
      const float ang = __fmul_rn(__fmul_rn(__fmul_rn(tf, wf[j]), 2.0f), 3.14159265358979323846f);
      f = k <= half ? sinf(ang) : cosf(ang);
    }
    feat[k] = f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = blockIdx.x * 16 + wave * 4 + i;
    if (c >= C) break;
    const float* wr = W + (int64_t)c * nfeat;
    float s = 0.f;
    for (int k = lane; k < nfeat; k += 64) s += feat[k] * wr[k];
    s = wave_sum64(s);
    if (lane == 0) {
      const float z = s + bias[c];
      const float sg = 1.0f / (1.0f + expf(-z));
      ds[(int64_t)b * C + c] = dcond[(int64_t)b * ldd + c] * (sg * (1.0f + z * (1.0f - sg)));  // silu'(z)
    }
  }
}

// Stage 2: dW[c, k] += sum_b ds[b, c] f[b, k]; dbias[c] += sum_b ds[b, c]  (one thread per (c, k); B is small)
__global__ __launch_bounds__(256) void time_cond_bwd_dw_kernel(const int32_t* __restrict__ times, int B, const float* __restrict__ wf, int half,
                                                               int C, const float* __restrict__ ds, float* __restrict__ dW,
                                                               float* __restrict__ dbias) {
  const int nfeat = 2 * half + 1;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)C * nfeat) return;
  const int c = (int)(i / nfeat), k = (int)(i - (int64_t)c * nfeat);
  float acc = 0.f, accb = 0.f;
  for (int b = 0; b < B; ++b) {
    const float tf = (float)times[b];
    float f = tf;
    if (k > 0) {
      const int j = k <= half ? k - 1 : k - 1 - half;
      const float ang = __fmul_rn(__fmul_rn(__fmul_rn(tf, wf[j]), 2.0f), 3.14159265358979323846f);
      f = k <= half ? sinf(ang) : cosf(ang);
    }
    const float d = ds[(int64_t)b * C + c];
    acc += d * f;
    accb += d;
  }
  dW[i] += acc;
  if (k == 0) dbias[c] += accb;
}

// Stage 3: dw_freq[j] += sum_b (df_sin[b, j] cos(ang) - df_cos[b, j] sin(ang)) * 2 pi t_b, df[b, k] = sum_c ds[b, c] W[c, k]
__global__ __launch_bounds__(256) void time_cond_bwd_freq_kernel(const int32_t* __restrict__ times, int B, const float* __restrict__ wf, int half,
                                                                 const float* __restrict__ W, int C, const float* __restrict__ ds,
                                                                 float* __restrict__ dwf) {
  const int j = blockIdx.x;  // one workgroup per frequency
  const int nfeat = 2 * half + 1;
  __shared__ float red[4];
  float total = 0.f;
  for (int b = 0; b < B; ++b) {
    float ps = 0.f, pc = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
      const float d = ds[(int64_t)b * C + c];
      ps += d * W[(int64_t)c * nfeat + 1 + j];
      pc += d * W[(int64_t)c * nfeat + 1 + half + j];
    }
    const float tf = (float)times[b];
    const float ang = __fmul_rn(__fmul_rn(__fmul_rn(tf, wf[j]), 2.0f), 3.14159265358979323846f);
    float v = (ps * cosf(ang) - pc * sinf(ang)) * (6.28318530717958647692f * tf);
    v = wave_sum64(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) total += (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
  }
  if (threadIdx.x == 0) dwf[j] += total;
}

}  // namespace dn

using namespace dn;

#define S_(stream) reinterpret_cast<hipStream_t>(stream)

extern "C" int dn_gate_forward(const void* h, const void* res, void* out, int32_t dtype, int32_t M, int32_t ld, int32_t T,
                               const float* gamma_beta, int32_t gb_ld, int32_t gb_half, void* stream) {
  DN_CHECK_ARG(h && res && out && M > 0 && ld > 0 && ld % 4 == 0 && T > 0, "dn_gate_forward: bad args");
  hipLaunchKernelGGL(gate_fwd_kernel, dim3(ew_blocks((int64_t)M * ld / 4)), dim3(256), 0, S_(stream), h, res, out, dtype, M, ld, T,
                     gamma_beta, gb_ld, gb_half);
  DN_CHECK_LAUNCH("dn_gate_forward");
  return DN_OK;
}

extern "C" int dn_gate_backward(const void* dout, const void* h, void* dh, int32_t dtype, int32_t M, int32_t ld, int32_t T,
                                const float* gamma_beta, int32_t gb_ld, int32_t gb_half, float* dgb_rows, int32_t dgb_ld, void* stream) {
  DN_CHECK_ARG(dout && h && dh && M > 0 && ld > 0 && ld % 4 == 0 && T > 0, "dn_gate_backward: bad args");
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(ew_blocks((int64_t)M * ld / 4)), dim3(256), 0, S_(stream), dout, h, dh, dtype, M, ld, T,
                     gamma_beta, gb_ld, gb_half, dgb_rows, dgb_ld);
  DN_CHECK_LAUNCH("dn_gate_backward");
  return DN_OK;
}

extern "C" int dn_geglu_forward(const void* pre, void* out, int32_t dtype, int32_t M, int32_t ip, void* stream) {
  DN_CHECK_ARG(pre && out && M > 0 && ip > 0 && ip % 8 == 0, "dn_geglu_forward: bad args");
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(ew_blocks((int64_t)M * ip / 4)), dim3(256), 0, S_(stream), pre, out, dtype, M, ip);
  DN_CHECK_LAUNCH("dn_geglu_forward");
  return DN_OK;
}

extern "C" int dn_geglu_backward(const void* dout, const void* pre, void* dpre, int32_t dtype, int32_t M, int32_t ip, void* stream) {
  DN_CHECK_ARG(dout && pre && dpre && M > 0 && ip > 0 && ip % 8 == 0, "dn_geglu_backward: bad args");
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(ew_blocks((int64_t)M * ip / 4)), dim3(256), 0, S_(stream), dout, pre, dpre, dtype, M, ip);
  DN_CHECK_LAUNCH("dn_geglu_backward");
  return DN_OK;
}

// rows handled by one workgroup of the norm backward: enough workgroups for the chip, partial rows kept small
static inline int norm_rows_per_block(int B, int T) {
  int rpb = 64;
  while (rpb > 8 && (int64_t)B * ((T + rpb - 1) / rpb) < 512) rpb >>= 1;
  return rpb;
}

extern "C" size_t dn_rmsnorm_backward_scratch_bytes(int32_t B, int32_t T, int32_t D) {
  const int rpb = norm_rows_per_block(B, T);
  return (size_t)B * ((T + rpb - 1) / rpb) * 2 * ((D + 63) / 64 * 64) * sizeof(float);
}

extern "C" int dn_rmsnorm_backward(const float* x, int32_t ldx, const void* dy, int32_t lddy, int32_t dy_dtype, int32_t B, int32_t T,
                                   int32_t D, const float* gamma, const float* gamma_beta, int32_t gb_ld, int32_t gb_half,
                                   const float* dres, float* dx, void* dx_act, int32_t act_dtype, int32_t ld_act, float* dgamma,
                                   float* dgamma_beta, int32_t dgb_ld, float* scratch, void* stream) {
  DN_CHECK_ARG(x && dy && dx && B > 0 && T > 0 && D > 0 && D <= 1024 && D % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0,
               "dn_rmsnorm_backward: bad args (D=%d must be a multiple of 4, <= 1024)", D);
  DN_CHECK_ARG(!(gamma && gamma_beta), "dn_rmsnorm_backward: learned and adaptive scale are exclusive");
  DN_CHECK_ARG((!dgamma && !dgamma_beta) || scratch, "dn_rmsnorm_backward: parameter gradients need the scratch buffer");
  const int rpb = norm_rows_per_block(B, T), bps = (T + rpb - 1) / rpb;
  const int part_ld = 2 * ((D + 63) / 64 * 64);
  const bool want = dgamma || dgamma_beta;
  hipLaunchKernelGGL(rmsnorm_bwd_kernel, dim3(B * bps), dim3(256), 0, S_(stream), x, ldx, dy, lddy, dy_dtype, B, T, D, gamma, gamma_beta,
                     gb_ld, part_ld / 2, dres, dx, dx_act, act_dtype, ld_act, rpb, bps, want ? scratch : nullptr, part_ld);
  if (dgamma)  // learned gamma: all workgroups reduce into one row
    hipLaunchKernelGGL(colsum_final_kernel, dim3((D + 63) / 64, 1), dim3(1024), 0, S_(stream), scratch, part_ld, B * bps, D, dgamma, 0,
                       1.0f, 1);
  if (dgamma_beta) {  // adaptive: per-sample rows [d g_c | d b_c]
    hipLaunchKernelGGL(colsum_final_kernel, dim3((D + 63) / 64, B), dim3(1024), 0, S_(stream), scratch, part_ld, bps, D, dgamma_beta,
                       dgb_ld, 1.0f, 1);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((D + 63) / 64, B), dim3(1024), 0, S_(stream), scratch + part_ld / 2, part_ld, bps, D,
                       dgamma_beta + gb_half, dgb_ld, 1.0f, 1);
  }
  DN_CHECK_LAUNCH("dn_rmsnorm_backward");
  return DN_OK;
}

// row blocks per group: about 64 rows each (8 per row lane), at most 1 Mi floats of partial sums in all (the engines' scratch)
static inline int colsum_chunks(int groups, int rows_per_group, int C) {
  const int64_t c4 = (C + 3) / 4 * 4;
  int chunks = (rows_per_group + 63) / 64;
  while (chunks > 1 && (int64_t)groups * chunks * c4 > (1 << 20)) chunks = (chunks + 1) / 2;
  return chunks < 1 ? 1 : chunks;
}

extern "C" size_t dn_colsum_scratch_bytes(int32_t groups, int32_t rows_per_group, int32_t C) {
  return (size_t)groups * colsum_chunks(groups, rows_per_group, C) * ((C + 3) / 4 * 4) * sizeof(float);
}

extern "C" int dn_colsum(const void* src, int32_t ld, int32_t dtype, int32_t groups, int32_t rows_per_group, int32_t C, float* out,
                         int32_t out_ld, float scale, int32_t accumulate, float* scratch, void* stream) {
  DN_CHECK_ARG(src && out && scratch && groups > 0 && rows_per_group > 0 && C > 0 && C % 4 == 0 && ld % 4 == 0 && ld >= C,
               "dn_colsum: bad args (C=%d ld=%d must be multiples of 4)", C, ld);
  const int chunks = colsum_chunks(groups, rows_per_group, C);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((C + 255) / 256, groups * chunks), dim3(256), 0, S_(stream), src, ld, dtype,
                     rows_per_group, chunks, C, scratch, C);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 63) / 64, groups), dim3(1024), 0, S_(stream), scratch, C, chunks, C, out, out_ld,
                     scale, accumulate);
  DN_CHECK_LAUNCH("dn_colsum");
  return DN_OK;
}

extern "C" int dn_posterior_backward(const float* params, int32_t ldp, const float* noise, int32_t ldn, const float* dz, int32_t lddz,
                                     void* dparams, int32_t act_dtype, int32_t ldo, int32_t M, int32_t Z, int32_t T,
                                     const int32_t* lengths, float kl_weight, void* stream) {
  DN_CHECK_ARG(params && noise && dz && dparams && M > 0 && Z > 0 && T > 0 && ldp >= 2 * Z && ldn >= Z && lddz >= Z && ldo >= 2 * Z,
               "dn_posterior_backward: bad args");
  hipLaunchKernelGGL(posterior_bwd_kernel, dim3(ew_blocks((int64_t)M * ldo)), dim3(256), 0, S_(stream), params, ldp, noise, ldn, dz, lddz,
                     dparams, act_dtype, ldo, M, Z, T, lengths, kl_weight);
  DN_CHECK_LAUNCH("dn_posterior_backward");
  return DN_OK;
}

extern "C" int dn_lsce_loss_grad(const float* logits, int32_t ld, const int32_t* target, int32_t M, int32_t V, float epsilon,
                                 float grad_scale, float* rows, void* dlogits, int32_t act_dtype, int32_t ldd, void* stream) {
  DN_CHECK_ARG(logits && target && rows && M > 0 && V > 1 && V <= 1024 && ld >= V && (!dlogits || (ldd >= V && ldd <= 1024)),
               "dn_lsce_loss_grad: bad args (V=%d must be <= 1024)", V);
  hipLaunchKernelGGL(lsce_kernel, dim3((M + 3) / 4), dim3(256), 0, S_(stream), logits, ld, target, M, V, epsilon, grad_scale, rows, dlogits,
                     act_dtype, ldd);
  DN_CHECK_LAUNCH("dn_lsce_loss_grad");
  return DN_OK;
}

extern "C" int dn_masked_mse_grad(const float* pred, int32_t ldp, const float* target, int32_t ldt, int32_t M, int32_t C, int32_t T,
                                  const int32_t* lengths, float grad_scale, float* sq_rows, float* dpred, int32_t ldd,
                                  int32_t accumulate, void* dpred_act, int32_t act_dtype, int32_t ld_act, void* stream) {
  DN_CHECK_ARG(pred && target && M > 0 && C > 0 && C % 4 == 0 && ldp % 4 == 0 && ldt % 4 == 0 && T > 0 && (!dpred || ldd % 4 == 0) &&
                   (!dpred_act || ld_act % 4 == 0),
               "dn_masked_mse_grad: bad args");
  hipLaunchKernelGGL(masked_mse_kernel, dim3((M + 3) / 4), dim3(256), 0, S_(stream), pred, ldp, target, ldt, M, C, T, lengths, grad_scale,
                     sq_rows, dpred, ldd, accumulate, dpred_act, act_dtype, ld_act);
  DN_CHECK_LAUNCH("dn_masked_mse_grad");
  return DN_OK;
}

extern "C" int dn_vec_sum(const float* v, int64_t n, float* out, int32_t accumulate, float* scratch, void* stream) {
  DN_CHECK_ARG(v && out && scratch && n > 0, "dn_vec_sum: bad args");
  hipLaunchKernelGGL(vec_sum_partial_kernel, dim3(kVecSumBlocks), dim3(256), 0, S_(stream), v, n, scratch);
  hipLaunchKernelGGL(vec_sum_final_kernel, dim3(1), dim3(256), 0, S_(stream), scratch, out, accumulate);
  DN_CHECK_LAUNCH("dn_vec_sum");
  return DN_OK;
}

extern "C" int dn_sum_groups(const void* src, int64_t stride, int32_t count, void* dst, int32_t dtype, int64_t n, void* stream) {
  DN_CHECK_ARG(src && dst && count >= 1 && n > 0 && n % 4 == 0 && stride % 4 == 0, "dn_sum_groups: bad args");
  hipLaunchKernelGGL(sum_groups_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, S_(stream), src, stride, count, dst, dtype, n);
  DN_CHECK_LAUNCH("dn_sum_groups");
  return DN_OK;
}

extern "C" size_t dn_rows_times_weight_scratch_bytes(int32_t B, int32_t N, int32_t C) {
  const int slices = N >= 128 * 64 ? 128 : (N + 63) / 64;
  return (size_t)slices * (B < 32 ? B : 32) * C * sizeof(float);
}

// see include/diffnorm_hip.h
extern "C" int dn_rows_times_weight(const float* X, int32_t ldx, int32_t B, const float* W, int32_t ldw, int32_t N, int32_t C, float* out,
                                    float* scratch, void* stream) {
  DN_CHECK_ARG(X && W && out && scratch && B >= 1 && N >= 1 && C >= 4 && C % 4 == 0 && ldw % 4 == 0 && ldw >= C && ldx >= N,
               "dn_rows_times_weight: bad args (C and ldw multiples of 4; B=%d N=%d C=%d)", B, N, C);
  DN_CHECK_ARG(((uintptr_t)W & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)scratch & 15) == 0, "dn_rows_times_weight: 16-byte alignment");
  const int slices = N >= 128 * 64 ? 128 : (N + 63) / 64;      // about a slice per CU-half: 256 workgroups at C = 2048
  const int per = ((N + slices - 1) / slices + 63) / 64 * 64;  // whole 64-row chunks
  dim3 grid((C / 4 + 255) / 256, (N + per - 1) / per);
  for (int b0 = 0; b0 < B; b0 += 32) {  // more than 32 rows: one stream of the matrix per 32
    const int nb = B - b0 < 32 ? B - b0 : 32;
    const float* Xb = X + (int64_t)b0 * ldx;
    if (nb <= 16) hipLaunchKernelGGL(rows_times_weight_kernel<16>, grid, dim3(256), 0, S_(stream), Xb, ldx, nb, W, ldw, N, C, per, scratch);
    else hipLaunchKernelGGL(rows_times_weight_kernel<32>, grid, dim3(256), 0, S_(stream), Xb, ldx, nb, W, ldw, N, C, per, scratch);
    DN_CHECK_LAUNCH("dn_rows_times_weight");
    const int rc = dn_sum_groups(scratch, (int64_t)nb * C, (int)grid.y, out + (int64_t)b0 * C, DN_F32, (int64_t)nb * C, stream);
    if (rc != DN_OK) return rc;
  }
  return DN_OK;
}

extern "C" int dn_transpose_weights(const void* src, int32_t dtype, int32_t count, int64_t src_stride, int32_t R, int32_t Cc, void* dst,
                                    int64_t dst_stride, int32_t Rp, int32_t Cp, void* stream) {
  DN_CHECK_ARG(src && dst && count > 0 && R > 0 && Cc > 0 && Rp >= R && Cp >= Cc && src_stride >= (int64_t)R * Cc &&
                   dst_stride >= (int64_t)Rp * Cp,
               "dn_transpose_weights: bad args");
  dim3 grid((Cp + 63) / 64, (Rp + 63) / 64, count);
  if (dtype == DN_BF16)
    hipLaunchKernelGGL(transpose_mat_kernel<uint16_t>, grid, dim3(256), 0, S_(stream), (const uint16_t*)src, src_stride, R, Cc,
                       (uint16_t*)dst, dst_stride, Rp, Cp);
  else
    hipLaunchKernelGGL(transpose_mat_kernel<float>, grid, dim3(256), 0, S_(stream), (const float*)src, src_stride, R, Cc, (float*)dst,
                       dst_stride, Rp, Cp);
  DN_CHECK_LAUNCH("dn_transpose_weights");
  return DN_OK;
}

extern "C" int dn_wgrad_reduce(const float* part, int32_t slices, int32_t cout, int32_t n_total, int32_t rows_w, int32_t n_taps,
                               float* grad, int32_t Np, int32_t Kp, void* stream) {
  DN_CHECK_ARG(part && grad && slices > 0 && cout > 0 && cout <= Np && n_taps > 0 && Kp % 4 == 0 && Kp <= rows_w &&
                   n_total >= n_taps * rows_w && n_total % 4 == 0 && rows_w % 4 == 0,
               "dn_wgrad_reduce: bad args");
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ew_blocks((int64_t)n_taps * cout * (Kp / 4))), dim3(256), 0, S_(stream), part, slices,
                     cout, n_total, rows_w, n_taps, grad, Np, Kp);
  DN_CHECK_LAUNCH("dn_wgrad_reduce");
  return DN_OK;
}

extern "C" int dn_add_broadcast(const float* src, float* dst, int32_t C, int32_t ld, int32_t count, void* stream) {
  DN_CHECK_ARG(src && dst && C > 0 && count > 0 && ld >= C, "dn_add_broadcast: bad args");
  hipLaunchKernelGGL(add_broadcast_kernel, dim3((C + 255) / 256), dim3(256), 0, S_(stream), src, dst, C, ld, count);
  DN_CHECK_LAUNCH("dn_add_broadcast");
  return DN_OK;
}

extern "C" int dn_transpose_pad_f32(const float* src, int32_t ld, int32_t B, int32_t T, int32_t C, int32_t front, int32_t Tp, float* dst,
                                    int32_t rows, int32_t rows_total, int32_t row0, int32_t chunk, void* stream) {
  DN_CHECK_ARG(src && dst && B > 0 && T > 0 && C > 0 && ld >= C && front >= 0 && Tp >= T + front && rows >= C, "dn_transpose_pad_f32: bad args");
  const int64_t cols = (int64_t)B * Tp;
  DN_CHECK_ARG(row0 >= 0 && row0 + rows <= rows_total && chunk > 0 && chunk % 32 == 0 && cols % chunk == 0, "dn_transpose_pad_f32: bad slices");
  dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64));
  hipLaunchKernelGGL(transpose_pad_f32_kernel, grid, dim3(256), 0, S_(stream), src, ld, B, T, C, front, Tp, dst, rows, rows_total, row0, chunk);
  DN_CHECK_LAUNCH("dn_transpose_pad_f32");
  return DN_OK;
}

extern "C" int dn_time_cond_backward(const int32_t* times, int32_t B, const float* w_freq, int32_t half, const float* W, const float* bias,
                                     int32_t C, const float* dcond, int32_t ldd, float* ds_scratch, float* dw_freq, float* dW, float* dbias,
                                     void* stream) {
  DN_CHECK_ARG(times && w_freq && W && bias && dcond && ds_scratch && dw_freq && dW && dbias && B > 0 && half > 0 && C > 0 && ldd >= C,
               "dn_time_cond_backward: bad args");
  const int nfeat = 2 * half + 1;
  hipLaunchKernelGGL(time_cond_bwd_ds_kernel, dim3((C + 15) / 16, B), dim3(256), nfeat * sizeof(float), S_(stream), times, w_freq, half, W, bias, C,
                     dcond, ldd, ds_scratch);
  hipLaunchKernelGGL(time_cond_bwd_dw_kernel, dim3((unsigned)(((int64_t)C * nfeat + 255) / 256)), dim3(256), 0, S_(stream), times, B, w_freq, half,
                     C, ds_scratch, dW, dbias);
  hipLaunchKernelGGL(time_cond_bwd_freq_kernel, dim3(half), dim3(256), 0, S_(stream), times, B, w_freq, half, W, C, ds_scratch, dw_freq);
  DN_CHECK_LAUNCH("dn_time_cond_backward");
  return DN_OK;
}
