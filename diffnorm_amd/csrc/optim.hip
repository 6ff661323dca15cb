// SURVEY 8 (f2), first piece: the optimizer step of the reference's training recipe (scripts/diffusion/train.sh:29-31:
// --optimizer adam --adam-betas '(0.9,0.98)' --clip-norm 2.0) over one flat fp32 master buffer.  HBM-bound: the
// gradient-norm pass reads 4 B per element, the update reads 16 B and writes 12 B (+2 B for the bf16 working copy).
//   clip:  fairseq/utils.py:347-397  (total_norm = ||g||_2 over all gradients; g *= min(1, max_norm / (total_norm + 1e-6)))
//   Adam:  fairseq/optim/adam.py:159-239  (denominator sqrt(v) + eps, bias corrections folded into the step size,
//          weight decay p += -wd * lr * p before the update)
// Both kernels take the norm from device memory, so norm -> clip -> update is three launches without a host round trip.
#include "common.h"

#include <algorithm>
#include <cmath>

namespace dn {

constexpr int kSumsqBlocks = 1024;  // fixed, so the two-level sum has one order whatever n is (deterministic)

__global__ __launch_bounds__(256) void grad_sumsq_partial_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four independent chains: the loop is load-latency-, not add-latency-bound
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = *reinterpret_cast<const float4*>(g + 4 * i);
    s0 += v.x * v.x; s1 += v.y * v.y; s2 += v.z * v.z; s3 += v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {
    const float v = g[(n4 << 2) + threadIdx.x];
    s0 += v * v;
  }
  float s = (s0 + s1) + (s2 + s3);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  __shared__ float ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// one workgroup: sumsq[0] (+)= sum of the kSumsqBlocks partials, in a fixed order
__global__ __launch_bounds__(256) void grad_sumsq_final_kernel(const float* __restrict__ partial, float* __restrict__ sumsq, int accumulate) {
  float s = 0.f;
  for (int i = threadIdx.x; i < kSumsqBlocks; i += 256) s += partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  __shared__ float ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = (ws[0] + ws[1]) + (ws[2] + ws[3]);
    sumsq[0] = accumulate ? sumsq[0] + t : t;
  }
}

struct AdamScalars {
  float beta1, beta2, one_m_beta1, one_m_beta2, eps, neg_step_size, neg_wd_lr, max_norm, grad_scale;
  const float* grad_scale_dev;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamScalars& h, float clip, float gs) {
  g *= gs;                                             // trainer.py:918-933  multiply_grads(world / sample_size)
  g *= clip;                                           // utils.py:394-396  g.mul_(clip_coef)
  m = __fmaf_rn(g, h.one_m_beta1, m * h.beta1);        // adam.py:215  exp_avg.mul_(beta1).add_(grad, alpha=1-beta1)
  v = __fmaf_rn(h.one_m_beta2, g * g, v * h.beta2);    // adam.py:216  exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
  const float denom = sqrtf(v) + h.eps;                // adam.py:223
  if (h.neg_wd_lr != 0.f) p = __fmaf_rn(p, h.neg_wd_lr, p);  // adam.py:229-232
  p = __fmaf_rn(h.neg_step_size, m / denom, p);        // adam.py:234  addcdiv_(exp_avg, denom, value=-step_size)
}

__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int64_t n, AdamScalars h,
                                                        const float* __restrict__ sumsq, uint16_t* __restrict__ p_bf16) {
  float clip = 1.f;
  const float gs = h.grad_scale_dev ? h.grad_scale * h.grad_scale_dev[0] : h.grad_scale;
  if (sumsq && h.max_norm > 0.f) clip = fminf(h.max_norm / (gs * sqrtf(sumsq[0]) + 1e-6f), 1.f);  // utils.py:392-394
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  auto update = [&](int64_t i, float4 pp, const float4 gg, float4 mm, float4 vv) {
    adam_one(pp.x, gg.x, mm.x, vv.x, h, clip, gs);
    adam_one(pp.y, gg.y, mm.y, vv.y, h, clip, gs);
    adam_one(pp.z, gg.z, mm.z, vv.z, h, clip, gs);
    adam_one(pp.w, gg.w, mm.w, vv.w, h, clip, gs);
    *reinterpret_cast<float4*>(p + 4 * i) = pp;
    *reinterpret_cast<float4*>(m + 4 * i) = mm;
    *reinterpret_cast<float4*>(v + 4 * i) = vv;
    if (p_bf16) *reinterpret_cast<uint2*>(p_bf16 + 4 * i) = make_uint2(pack_bf16x2(pp.x, pp.y), pack_bf16x2(pp.z, pp.w));
  };
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {  // two elements' eight 16-byte loads in flight per lane before the first use
    const int64_t j = i + stride;
    const float4 p0 = *reinterpret_cast<const float4*>(p + 4 * i), g0 = *reinterpret_cast<const float4*>(g + 4 * i);
    const float4 m0 = *reinterpret_cast<const float4*>(m + 4 * i), v0 = *reinterpret_cast<const float4*>(v + 4 * i);
    const float4 p1 = *reinterpret_cast<const float4*>(p + 4 * j), g1 = *reinterpret_cast<const float4*>(g + 4 * j);
    const float4 m1 = *reinterpret_cast<const float4*>(m + 4 * j), v1 = *reinterpret_cast<const float4*>(v + 4 * j);
    update(i, p0, g0, m0, v0);
    update(j, p1, g1, m1, v1);
  }
  if (i < n4)
    update(i, *reinterpret_cast<const float4*>(p + 4 * i), *reinterpret_cast<const float4*>(g + 4 * i),
           *reinterpret_cast<const float4*>(m + 4 * i), *reinterpret_cast<const float4*>(v + 4 * i));
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    float pp = p[i], mm = m[i], vv = v[i];
    adam_one(pp, g[i], mm, vv, h, clip, gs);
    p[i] = pp; m[i] = mm; v[i] = vv;
    if (p_bf16) p_bf16[i] = (uint16_t)pack_bf16x2(pp, 0.f);
  }
}

// ---- frames-major -> channels-major with per-sequence front padding (operand of the weight-gradient contraction) ----
// dst[c, b * Tp + front + t] = src[b * T + t, c]; every other element of dst [rows, B * Tp] (pad columns, rows >= C) = 0.
// The B * Tp columns are stored in groups of `chunk`: dst is [cols / chunk][rows][chunk], one K-slice of a split-K contraction
// per group with its rows contiguous (the contraction's weight operand has no row stride of its own); this call fills rows
// [row0, row0 + rows) of the destination's rows_total (the taps of a conv stack their transposed inputs in one operand).
// 64 x 64 tiles through LDS: 128-byte reads along c, 128-byte writes along the frame index.
__global__ __launch_bounds__(256) void transpose_pad_kernel(const uint16_t* __restrict__ src, int ld, int B, int T, int C, int front, int Tp,
                                                            uint16_t* __restrict__ dst, int rows, int rows_total, int row0,
                                                            int chunk) {
  __shared__ uint16_t tile[64][66];
  const int64_t cols = (int64_t)B * Tp;
  const int64_t j0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {  // r: frame within the tile, tx: channel
    const int64_t j = j0 + r;
    uint16_t v = 0;
    if (j < cols) {
      const int b = (int)(j / Tp), t = (int)(j - (int64_t)b * Tp) - front;
      const int c = c0 + tx;
      if (t >= 0 && t < T && c < C) v = src[((int64_t)b * T + t) * ld + c];
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {  // r: channel within the tile, tx: frame
    const int c = c0 + r;
    const int64_t j = j0 + tx;
    if (c < rows && j < cols) dst[((j / chunk) * rows_total + row0 + c) * chunk + j % chunk] = tile[tx][r];
  }
}

}  // namespace dn

extern "C" int dn_transpose_pad(const void* src, int32_t ld, int32_t B, int32_t T, int32_t C, int32_t front, int32_t Tp, void* dst,
                                int32_t rows, int32_t rows_total, int32_t row0, int32_t chunk, void* stream) {
  DN_CHECK_ARG(src && dst && B > 0 && T > 0 && C > 0 && ld >= C && front >= 0 && Tp >= T + front && rows >= C,
               "dn_transpose_pad: B=%d T=%d C=%d ld=%d front=%d Tp=%d rows=%d", B, T, C, ld, front, Tp, rows);
  const int64_t cols = (int64_t)B * Tp;
  DN_CHECK_ARG(row0 >= 0 && row0 + rows <= rows_total, "dn_transpose_pad: rows [%d, %d) outside the destination's %d", row0, row0 + rows, rows_total);
  DN_CHECK_ARG(chunk > 0 && chunk % 64 == 0 && cols % chunk == 0, "dn_transpose_pad: chunk=%d must be a multiple of 64 dividing B*Tp=%lld",
               chunk, (long long)cols);
  dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64));
  hipLaunchKernelGGL(dn::transpose_pad_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const uint16_t*>(src), ld, B, T, C, front, Tp, reinterpret_cast<uint16_t*>(dst), rows, rows_total, row0, chunk);
  DN_CHECK_LAUNCH("dn_transpose_pad");
  return DN_OK;
}

extern "C" int dn_grad_sumsq(const float* grad, int64_t n, float* scratch, float* sumsq, int32_t accumulate, void* stream) {
  DN_CHECK_ARG(grad && scratch && sumsq && n > 0, "dn_grad_sumsq: null pointer or n=%lld", (long long)n);
  DN_CHECK_ARG((reinterpret_cast<uintptr_t>(grad) & 15) == 0, "dn_grad_sumsq: grad must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(dn::grad_sumsq_partial_kernel, dim3(dn::kSumsqBlocks), dim3(256), 0, s, grad, n, scratch);
  hipLaunchKernelGGL(dn::grad_sumsq_final_kernel, dim3(1), dim3(256), 0, s, scratch, sumsq, accumulate);
  DN_CHECK_LAUNCH("dn_grad_sumsq");
  return DN_OK;
}

extern "C" int dn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const DnAdamParams* hp,
                            const float* sumsq, void* param_bf16, void* stream) {
  DN_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && hp && n > 0, "dn_adam_step: null pointer or n=%lld", (long long)n);
  DN_CHECK_ARG(hp->step >= 1, "dn_adam_step: step=%d (the first update is step 1)", hp->step);
  DN_CHECK_ARG(hp->beta1 >= 0. && hp->beta1 < 1. && hp->beta2 >= 0. && hp->beta2 < 1. && hp->eps >= 0.,
               "dn_adam_step: betas (%g, %g) eps %g", hp->beta1, hp->beta2, hp->eps);
  for (const void* q : {(const void*)param, (const void*)grad, (const void*)exp_avg, (const void*)exp_avg_sq})
    DN_CHECK_ARG((reinterpret_cast<uintptr_t>(q) & 15) == 0, "dn_adam_step: buffers must be 16-byte aligned");
  DN_CHECK_ARG((reinterpret_cast<uintptr_t>(param_bf16) & 7) == 0, "dn_adam_step: param_bf16 must be 8-byte aligned");
  // adam.py:225-227, in double like the reference's Python floats
  const double bc1 = 1.0 - pow(hp->beta1, hp->step), bc2 = 1.0 - pow(hp->beta2, hp->step);
  const double step_size = hp->lr * sqrt(bc2) / bc1;
  dn::AdamScalars h;
  h.beta1 = (float)hp->beta1; h.beta2 = (float)hp->beta2;
  h.one_m_beta1 = (float)(1.0 - hp->beta1); h.one_m_beta2 = (float)(1.0 - hp->beta2);
  h.eps = (float)hp->eps; h.neg_step_size = (float)-step_size;
  h.neg_wd_lr = hp->weight_decay != 0. ? (float)(-hp->weight_decay * hp->lr) : 0.f;
  h.max_norm = (float)hp->max_norm;
  h.grad_scale = hp->grad_scale != 0. ? (float)hp->grad_scale : 1.f;
  h.grad_scale_dev = hp->grad_scale_dev;
  const int64_t n4 = (n + 3) >> 2;
  static const int wg_per_cu = getenv("DN_ADAM_WG_PER_CU") ? atoi(getenv("DN_ADAM_WG_PER_CU")) : 8;  // measured best of 4..32 at 384 Mi elements
  const int blocks = (int)std::min<int64_t>((n4 + 255) / 256, 256 * wg_per_cu);  // grid-stride beyond
  hipLaunchKernelGGL(dn::adam_step_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), param, grad, exp_avg,
                     exp_avg_sq, n, h, sumsq, reinterpret_cast<uint16_t*>(param_bf16));
  DN_CHECK_LAUNCH("dn_adam_step");
  return DN_OK;
}
