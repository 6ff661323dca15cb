// Kernels shared by the NAR decoder and encoder engines (nar_decoder.hip, nar_encoder.hip): internal linkage, one copy per unit.
#pragma once
#include "common.h"

namespace dn {
namespace {

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// LayerNorm (eps 1e-5, biased variance, affine): x fp32 [M, ldx] -> y [M, ldy] in out_dtype (pad columns zeroed).  One wave per
// row, D <= 1024 in registers.
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx, void* __restrict__ y, int ldy, int out_dtype, int M,
                                                        int D, const float* __restrict__ gamma, const float* __restrict__ beta) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (int64_t)row * ldx;
  float4 v[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (i * 64 + lane) * 4;
    v[i] = c < D ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0, 0, 0, 0);
    s += v[i].x + v[i].y + v[i].z + v[i].w;
  }
  const float mean = wave_sum64(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < D) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += a * a + b * b + cc * cc + d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum64(q) / (float)D + 1e-5f);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c >= ldy) continue;
    float o[4] = {0, 0, 0, 0};
    if (c < D) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
      o[0] = (v[i].x - mean) * rstd * g.x + be.x; o[1] = (v[i].y - mean) * rstd * g.y + be.y;
      o[2] = (v[i].z - mean) * rstd * g.z + be.z; o[3] = (v[i].w - mean) * rstd * g.w + be.w;
    }
    store4(y, (int64_t)row * ldy + c, out_dtype, o[0], o[1], o[2], o[3]);
  }
}


}  // namespace
}  // namespace dn
