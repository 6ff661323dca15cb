// HBM-bound kernels of the path: norms, timestep conditioning, scheduler updates, posterior sampling,
// argmax, Philox normal fill, row conversion.  All are one pass over their operands with 16-byte
// accesses where the layout allows; reductions are wave-level (64 lanes) with DPP-free shuffles.
#include "common.h"

namespace dn {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ------------------------------------------------------------------------------------------ RMSNorm
// one wave per row; D <= 1024 keeps the row in registers (4 float4 per lane), larger D re-reads.
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, int ldx, void* __restrict__ y, int ldy,
                                                      int out_dtype, int M, int D, int T, const float* __restrict__ gamma,
                                                      const float* __restrict__ gb, int gb_ld, int gb_half) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (int64_t)row * ldx;
  float4 v[4];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (i * 64 + lane) * 4;
    v[i] = c < D ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0, 0, 0, 0);
    ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
  }
  for (int c = (256 + lane) * 4; c < D; c += 256) {
    const float4 u = *reinterpret_cast<const float4*>(xr + c);
    ss += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w;
  }
  ss = wave_sum(ss);
  const float denom = fmaxf(sqrtf(ss), 1e-12f);
  const float scale = sqrtf((float)D);
  const float* gbr = gb ? gb + (int64_t)(row / T) * gb_ld : nullptr;
  auto emit = [&](int c, float4 u) {
    float o[4] = {u.x / denom * scale, u.y / denom * scale, u.z / denom * scale, u.w / denom * scale};
    if (gamma) {
      const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
      o[0] *= ga.x; o[1] *= ga.y; o[2] *= ga.z; o[3] *= ga.w;
    }
    if (gbr) {
      const float4 ga = *reinterpret_cast<const float4*>(gbr + c);
      const float4 be = *reinterpret_cast<const float4*>(gbr + gb_half + c);
      o[0] = o[0] * ga.x + be.x; o[1] = o[1] * ga.y + be.y; o[2] = o[2] * ga.z + be.z; o[3] = o[3] * ga.w + be.w;
    }
    store4(y, (int64_t)row * ldy + c, out_dtype, o[0], o[1], o[2], o[3]);
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < D) emit(c, v[i]);
  }
  for (int c = (256 + lane) * 4; c < D; c += 256) emit(c, *reinterpret_cast<const float4*>(xr + c));
  // zero the pad columns [D, ldy) so the row can feed a K-padded contraction
  for (int c = D + lane * 4; c < ldy; c += 256) store4(y, (int64_t)row * ldy + c, out_dtype, 0.f, 0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------------------------------ time cond
// block = (16 outputs, one sample): the 2*half+1 Fourier features go to LDS once, then each wave does
// 4 dot products of length 2*half+1.  The angle is formed exactly as the reference forms it in fp32:
// fl(fl(fl(t*w)*2)*pi_f32) -- t up to 999 and w ~ N(0,1) give arguments of 1e3..1e4 rad, so sinf/cosf
// (full range reduction) are used, never the fast intrinsics.
__global__ __launch_bounds__(256) void time_cond_kernel(const int32_t* __restrict__ times, const float* __restrict__ wf, int half,
                                                        const float* __restrict__ W, const float* __restrict__ bias, int C,
                                                        float* __restrict__ out, void* __restrict__ out_act, int act_dtype, int ldo) {
  extern __shared__ float feat[];
  const int b = blockIdx.y;
  const int nfeat = 2 * half + 1;
  const float tf = (float)times[b];
  for (int k = threadIdx.x; k < nfeat; k += 256) {
    float f;
    if (k == 0) {
      f = tf;
    } else {
      const int j = k <= half ? k - 1 : k - 1 - half;
      float ang = __fmul_rn(__fmul_rn(__fmul_rn(tf, wf[j]), 2.0f), 3.14159265358979323846f);
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
    s = wave_sum(s);
    if (lane == 0) {
      const float v = silu(s + bias[c]);
      out[(int64_t)b * ldo + c] = v;
      if (out_act) {
        if (act_dtype == DN_BF16X3)
          store1_split(out_act, (int64_t)b * ldo + c, v);
        else if (dn_is16(act_dtype))
          reinterpret_cast<uint16_t*>(out_act)[(int64_t)b * ldo + c] = to_h16(act_dtype, v);
        else
          reinterpret_cast<float*>(out_act)[(int64_t)b * ldo + c] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ scheduler
__device__ __forceinline__ void store_act1(void* p, int64_t off, int dtype, float v) {
  if (dtype == DN_BF16X3)
    store1_split(p, off, v);
  else if (dn_is16(dtype))
    reinterpret_cast<uint16_t*>(p)[off] = to_h16(dtype, v);
  else
    reinterpret_cast<float*>(p)[off] = v;
}

__global__ __launch_bounds__(256) void ddim_step_kernel(const float* x, const float* __restrict__ eps, float* xo,
                                                        void* __restrict__ xact, int act_dtype, int ld_act, int M, int C, int ld,
                                                        int T, const float* __restrict__ coef, const int32_t* __restrict__ t) {
  const int64_t n = (int64_t)M * C;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int m = (int)(i / C), c = (int)(i - (int64_t)m * C);
    const float* cf = coef + 4 * t[m / T];
    const float sa = cf[0], s1 = cf[1];
    const int64_t o = (int64_t)m * ld + c;
    const float xv = x[o], ev = eps[o];
    // x1_hat = safe_div(x - s1*eps, sa); pred_noise = safe_div(x - sa*x1_hat, s1); mean = x1_hat*sqrt(abp) + sqrt(1-abp)*pn
    const float x1 = __fdiv_rn(__fsub_rn(xv, __fmul_rn(s1, ev)), fmaxf(sa, 1e-10f));
    const float pn = __fdiv_rn(__fsub_rn(xv, __fmul_rn(sa, x1)), fmaxf(s1, 1e-10f));
    const float r = __fadd_rn(__fmul_rn(x1, cf[2]), __fmul_rn(cf[3], pn));
    xo[o] = r;
    if (xact) store_act1(xact, (int64_t)m * ld_act + c, act_dtype, r);
  }
}

__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x, const float* __restrict__ noise, float* __restrict__ out,
                                                       void* __restrict__ oact, int act_dtype, int ld_act, int M, int C, int ld, int T,
                                                       const float* __restrict__ ca, const float* __restrict__ cb,
                                                       const int32_t* __restrict__ t) {
  const int64_t n = (int64_t)M * C;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int m = (int)(i / C), c = (int)(i - (int64_t)m * C);
    const int tb = t[m / T];
    const int64_t o = (int64_t)m * ld + c;
    const float r = __fadd_rn(__fmul_rn(ca[tb], x[o]), __fmul_rn(cb[tb], noise[o]));
    out[o] = r;
    if (oact) store_act1(oact, (int64_t)m * ld_act + c, act_dtype, r);
  }
}

// ------------------------------------------------------------------------------------------ posterior
__global__ __launch_bounds__(256) void posterior_kernel(const float* __restrict__ params, int ldp, const float* __restrict__ noise, int ldn,
                                                        float* __restrict__ z, void* __restrict__ zact, int act_dtype, int ldz, int M,
                                                        int Z, int T, const int32_t* __restrict__ lengths, float* __restrict__ kl_rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* pr = params + (int64_t)row * ldp;
  float kl = 0.f;
  for (int c = lane; c < Z; c += 64) {
    const float mean = pr[c];
    const float lv = fminf(fmaxf(pr[Z + c], -30.0f), 20.0f);
    const float sd = expf(0.5f * lv);
    const float v = __fadd_rn(mean, __fmul_rn(sd, noise[(int64_t)row * ldn + c]));
    z[(int64_t)row * ldz + c] = v;
    if (zact) store_act1(zact, (int64_t)row * ldz + c, act_dtype, v);
    kl += mean * mean + expf(lv) - 1.0f - lv;
  }
  for (int c = Z + lane; c < ldz; c += 64) {
    z[(int64_t)row * ldz + c] = 0.f;
    if (zact) store_act1(zact, (int64_t)row * ldz + c, act_dtype, 0.f);
  }
  if (kl_rows) {
    kl = wave_sum(kl);
    const int b = row / T, t = row - b * T;
    const bool valid = lengths ? t < lengths[b] : true;
    if (lane == 0) kl_rows[row] = valid ? 0.5f * kl : 0.f;
  }
}

// ------------------------------------------------------------------------------------------ argmax
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, int ld, int M, int V, int offset,
                                                     int32_t* __restrict__ units) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* lr = logits + (int64_t)row * ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < V; c += 64) {
    const float v = lr[c];
    if (v > best || (v == best && c < bi)) {  // first maximum wins, like torch.argmax
      best = v;
      bi = c;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  if (lane == 0) units[row] = bi - offset;
}

// ------------------------------------------------------------------------------------------ Philox randn
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  c[1] = (uint32_t)p1;
  c[3] = (uint32_t)p0;
  c[0] = n0;
  c[2] = n2;
}

__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset) {
  const int64_t nquad = (n + 3) >> 2;
  for (int64_t q = blockIdx.x * 256 + threadIdx.x; q < nquad; q += (int64_t)gridDim.x * 256) {
    const uint64_t ctr = offset + (uint64_t)q;
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    float r[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
      const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const float rad = sqrtf(-2.0f * logf(u1));
      float sn, cs;
      sincosf(6.28318530717958647692f * u2, &sn, &cs);
      r[2 * h] = rad * cs;
      r[2 * h + 1] = rad * sn;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (q * 4 + j < n) out[q * 4 + j] = r[j];
  }
}

// ------------------------------------------------------------------------------------------ ancestral (DDPM) step of the device loop
// p_sample (diffusion/gaussian_diffusion.py:376-417) with a fixed variance, the arithmetic of gaussian_step_kernel's sampler 0;
// the noise is injected (`noise`, the row of this step) or drawn here: Philox4x32-10, key = seed, counter = (element quad, t) --
// t comes from the per-sample step vector the loop fills from its device counter, so graph replays draw fresh noise.
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t ctr_lo, uint32_t ctr_hi, float (&r)[4]);
__global__ __launch_bounds__(256) void ddpm_step_kernel(float* x, const float* __restrict__ eps, int M, int C, int T,
                                                        const float* __restrict__ table, const int32_t* __restrict__ t,
                                                        int clip, const float* __restrict__ noise, int64_t noise_row, int t_top,
                                                        uint64_t seed) {
  const int64_t nquad = ((int64_t)M * C) >> 2;  // C is a multiple of 4: a quad stays inside one row
  for (int64_t q = blockIdx.x * 256 + threadIdx.x; q < nquad; q += (int64_t)gridDim.x * 256) {
    const int64_t i = q << 2;
    const int m = (int)(i / C);
    const int tn = t[m / T];
    const float* tb = table + (int64_t)tn * DN_GD_COLS;
    const float4 xv = *reinterpret_cast<const float4*>(x + i), ev = *reinterpret_cast<const float4*>(eps + i);
    float z[4];
    if (noise) {
      const float4 nv = *reinterpret_cast<const float4*>(noise + (int64_t)(t_top - tn) * noise_row + i);
      z[0] = nv.x; z[1] = nv.y; z[2] = nv.z; z[3] = nv.w;
    } else {
      philox_normal4(seed, (uint64_t)q, (uint32_t)tn, z);
    }
    const float sd = tn != 0 ? expf(__fmul_rn(0.5f, tb[4])) : 0.0f;
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, es[4] = {ev.x, ev.y, ev.z, ev.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x0 = __fsub_rn(__fmul_rn(tb[0], xs[j]), __fmul_rn(tb[1], es[j]));
      if (clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
      const float mean = __fadd_rn(__fmul_rn(tb[2], x0), __fmul_rn(tb[3], xs[j]));
      o[j] = __fadd_rn(mean, __fmul_rn(sd, z[j]));
    }
    *reinterpret_cast<float4*>(x + i) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t ctr_lo, uint32_t ctr_hi, float (&r)[4]) {
  uint32_t c[4] = {(uint32_t)ctr_lo, (uint32_t)(ctr_lo >> 32), ctr_hi, 0x44504d50u};  // ("DPMP": a stream apart from dn_randn's)
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {  // Box-Muller on (0,1) uniforms, as randn_kernel
    const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.28318530717958647692f * u2, &sn, &cs);
    r[2 * h] = rad * cs;
    r[2 * h + 1] = rad * sn;
  }
}

// ------------------------------------------------------------------------------------------ convert rows
__global__ __launch_bounds__(256) void convert_rows_kernel(const void* __restrict__ src, int sdt, int lds, void* __restrict__ dst, int ddt,
                                                           int ldd, int M, int C) {
  const int64_t n = (int64_t)M * ldd;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int m = (int)(i / ldd), c = (int)(i - (int64_t)m * ldd);
    float v = 0.f;
    if (c < C) {
      const int64_t so = (int64_t)m * lds + c;
      v = sdt == DN_BF16X3 ? load1_split(src, so) : dn_is16(sdt) ? from_h16(sdt, reinterpret_cast<const uint16_t*>(src)[so]) : reinterpret_cast<const float*>(src)[so];
    }
    store_act1(dst, i, ddt, v);
  }
}


// ------------------------------------------------------------------------------------------ generic Gaussian step
// One reverse step of the OpenAI-style scheduler for an eps-predicting model: pred_xstart (optionally clipped),
// posterior mean, fixed / learned-range variance, then either the ancestral sample (p_sample) or the DDIM update.
// table: fp32 [T, DN_GD_COLS] = {sqrt_recip_abar, sqrt_recipm1_abar, post_coef1, post_coef2, fixed_log_var,
// min_log, max_log, abar, abar_prev} -- the fp32 casts of the float64 schedule, as _extract_into_tensor makes.
__global__ __launch_bounds__(256) void gaussian_step_kernel(const DnGaussianStep p) {
  const int64_t total = (int64_t)p.N * p.inner;
  const int cmul = p.learned_range ? 2 : 1;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i / p.inner);
    const int64_t rem = i - (int64_t)n * p.inner;
    const int tn = p.t[n];
    const float* tb = p.table + (int64_t)tn * DN_GD_COLS;
    const float xv = p.x[i];
    const float eps = p.model_out[(int64_t)n * p.inner * cmul + rem];
    float x0 = p.predict_xstart ? eps : __fsub_rn(__fmul_rn(tb[0], xv), __fmul_rn(tb[1], eps));  // START_X: the output is x_0 (:317-318)
    if (p.clip_denoised) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
    float logvar = tb[4], var = tb[11];
    if (p.learned_range) {
      const float v = p.model_out[(int64_t)n * p.inner * 2 + p.inner + rem];
      const float frac = __fdiv_rn(__fadd_rn(v, 1.0f), 2.0f);
      logvar = __fadd_rn(__fmul_rn(frac, tb[6]), __fmul_rn(__fsub_rn(1.0f, frac), tb[5]));
      var = expf(logvar);
    }
    const float nz = tn != 0 ? 1.0f : 0.0f;
    const float nv = p.noise ? p.noise[i] : 0.0f;
    float out;
    if (p.sampler == 0) {  // p_sample: mean + 1[t != 0] * exp(0.5 * log_variance) * noise
      float mean = __fadd_rn(__fmul_rn(tb[2], x0), __fmul_rn(tb[3], xv));
      if (p.cond_grad) mean = __fadd_rn(mean, __fmul_rn(var, p.cond_grad[i]));  // condition_mean (:346-358)
      out = __fadd_rn(mean, __fmul_rn(__fmul_rn(nz, expf(__fmul_rn(0.5f, logvar))), nv));
    } else {  // ddim_sample
      if (p.cond_grad) {  // condition_score (:360-374): shift eps by the conditional score, re-derive the x_0 prediction (not re-clipped)
        float ec = __fdiv_rn(__fsub_rn(__fmul_rn(tb[0], xv), x0), tb[1]);
        ec = __fsub_rn(ec, __fmul_rn(sqrtf(__fsub_rn(1.0f, tb[7])), p.cond_grad[i]));
        x0 = __fsub_rn(__fmul_rn(tb[0], xv), __fmul_rn(tb[1], ec));
      }
      const float e2 = __fdiv_rn(__fsub_rn(__fmul_rn(tb[0], xv), x0), tb[1]);
      const float ab = tb[7], abp = tb[8];
      const float sigma = __fmul_rn(__fmul_rn(p.eta, sqrtf(__fdiv_rn(__fsub_rn(1.0f, abp), __fsub_rn(1.0f, ab)))),
                                    sqrtf(__fsub_rn(1.0f, __fdiv_rn(ab, abp))));
      const float mean = __fadd_rn(__fmul_rn(x0, sqrtf(abp)),
                                   __fmul_rn(sqrtf(__fsub_rn(__fsub_rn(1.0f, abp), __fmul_rn(sigma, sigma))), e2));
      out = __fadd_rn(mean, __fmul_rn(__fmul_rn(nz, sigma), nv));
    }
    p.sample[i] = out;
    if (p.pred_xstart) p.pred_xstart[i] = x0;
  }
}

// p_mean_variance / q_posterior_mean_variance / ddim_reverse_sample / the VB term, one thread per element (header: DnGaussianMoments)
__global__ __launch_bounds__(256) void gaussian_moments_kernel(const DnGaussianMoments p) {
  const int64_t total = (int64_t)p.N * p.inner;
  const int cmul = p.learned_range ? 2 : 1;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i / p.inner);
    const int64_t rem = i - (int64_t)n * p.inner;
    const int tn = p.t[n];
    const float* tb = p.table + (int64_t)tn * DN_GD_COLS;
    const float xv = p.x[i];
    if (!p.model_out) {  // q(x_{t-1} | x_t, x_0)
      const float xs = p.x_start[i];
      if (p.mean) p.mean[i] = __fadd_rn(__fmul_rn(tb[2], xs), __fmul_rn(tb[3], xv));
      if (p.variance) p.variance[i] = tb[10];
      if (p.log_variance) p.log_variance[i] = tb[5];
      continue;
    }
    const float eps = p.model_out[(int64_t)n * p.inner * cmul + rem];
    float x0 = p.predict_xstart ? eps : __fsub_rn(__fmul_rn(tb[0], xv), __fmul_rn(tb[1], eps));
    if (p.clip_denoised) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
    float logvar = tb[4], var = tb[11];
    if (p.learned_range) {
      const float v = p.model_out[(int64_t)n * p.inner * 2 + p.inner + rem];
      const float frac = __fdiv_rn(__fadd_rn(v, 1.0f), 2.0f);
      logvar = __fadd_rn(__fmul_rn(frac, tb[6]), __fmul_rn(__fsub_rn(1.0f, frac), tb[5]));
      var = expf(logvar);
    }
    const float mean = __fadd_rn(__fmul_rn(tb[2], x0), __fmul_rn(tb[3], xv));
    if (p.mean) p.mean[i] = mean;
    if (p.variance) p.variance[i] = var;
    if (p.log_variance) p.log_variance[i] = logvar;
    if (p.pred_xstart) p.pred_xstart[i] = x0;
    if (p.reverse_sample) {  // (:583-596)
      const float e2 = __fdiv_rn(__fsub_rn(__fmul_rn(tb[0], xv), x0), tb[1]);
      const float abn = tb[9];
      p.reverse_sample[i] = __fadd_rn(__fmul_rn(x0, sqrtf(abn)), __fmul_rn(sqrtf(__fsub_rn(1.0f, abn)), e2));
    }
    if (p.vb && p.x_start) {
      const float xs = p.x_start[i];
      float v;
      if (tn != 0) {  // normal_kl(true_mean, true_log_variance_clipped, mean, log_variance), diffusion_utils.py:10-34
        const float tm = __fadd_rn(__fmul_rn(tb[2], xs), __fmul_rn(tb[3], xv));
        const float lv1 = tb[5], d = tm - mean;
        v = 0.5f * (-1.0f + logvar - lv1 + expf(lv1 - logvar) + d * d * expf(-logvar));
      } else {  // -discretized_gaussian_log_likelihood(x_start, means = mean, log_scales = 0.5 * log_variance)
        const float c = xs - mean, inv = expf(-0.5f * logvar);
        auto cdf = [](float z) { return 0.5f * (1.0f + tanhf(0.79788456080286535588f * (z + 0.044715f * z * z * z))); };
        const float cp = cdf(inv * (c + 1.0f / 255.0f)), cm = cdf(inv * (c - 1.0f / 255.0f));
        const float lp = xs < -0.999f ? logf(fmaxf(cp, 1e-12f)) : (xs > 0.999f ? logf(fmaxf(1.0f - cm, 1e-12f)) : logf(fmaxf(cp - cm, 1e-12f)));
        v = -lp;
      }
      p.vb[i] = v * 1.44269504088896340736f;  // / ln 2: bits
    }
  }
}

// ------------------------------------------------------------------------------------------ CMLM mask-predict update (f4)
// One workgroup per sequence (T <= 2048).  Phase 1: every position holding `unk` (the mask symbol) takes argmax / max of
// log_softmax(logits[b, t, :]) (one wave per position).  Phase 2 (not on the last iteration): the boundary lowest-scoring
// positions -- boundary = trunc((n_nonpad - 2) * p), scores ascending, ties by position -- are re-masked (token = unk, score = 0).
// step_dev != nullptr: the iteration index is read from the device (a captured refinement iteration is replayed while a device
// counter advances: nothing of the iteration is baked into the graph) and p / remask are derived here exactly as the host derives them.
__global__ __launch_bounds__(256) void cmlm_step_kernel(const float* __restrict__ logits, int32_t* __restrict__ tokens, float* __restrict__ scores,
                                                        int32_t* __restrict__ predicted, int T, int V, float p, int remask, int unk, int pad,
                                                        const int32_t* __restrict__ step_dev, int max_step) {
  if (step_dev) {
    const int step = *step_dev;
    remask = (step + 1) < max_step;
    p = (float)(1.0 - (double)(step + 1) / (double)max_step);  // the Python float of the reference, cast to the scores' fp32
  }
  __shared__ float s_sc[2048];
  __shared__ int32_t s_tok[2048];
  __shared__ int s_n;
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_n = 0;
  for (int t = wave; t < T; t += 4) {
    int32_t tok = tokens[(int64_t)b * T + t];
    float sc = scores[(int64_t)b * T + t];
    if (tok == unk) {  // wave-uniform
      const float* lr = logits + ((int64_t)b * T + t) * V;
      float best = -INFINITY;
      int bi = 0x7fffffff;
      for (int c = lane; c < V; c += 64) {
        const float v = lr[c];
        if (v > best) { best = v; bi = c; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      float se = 0.f;
      for (int c = lane; c < V; c += 64) se += expf(lr[c] - best);
      se = wave_sum(se);
      tok = bi;
      sc = -logf(se);  // max - logsumexp = -(log sum exp(x - max))
    }
    if (lane == 0) {
      s_tok[t] = tok;
      s_sc[t] = sc;
      predicted[(int64_t)b * T + t] = tok;
    }
  }
  __syncthreads();
  if (remask) {
    int cnt = 0;
    for (int t = threadIdx.x; t < T; t += 256) cnt += s_tok[t] != pad;
    if (cnt) atomicAdd(&s_n, cnt);  // integer count within the workgroup: order-independent
    __syncthreads();
    const long long boundary = (long long)((float)(s_n - 2) * p);
    for (int i = threadIdx.x; i < T; i += 256) {
      const float si = s_sc[i];
      int rank = 0;
      for (int j = 0; j < T; ++j) {
        const float sj = s_sc[j];
        rank += (sj < si) || (sj == si && j < i);
      }
      int32_t tok = s_tok[i];
      float sc = si;
      if ((long long)rank < boundary) { tok = unk; sc = 0.f; }
      tokens[(int64_t)b * T + i] = tok;
      scores[(int64_t)b * T + i] = sc;
    }
  } else {
    for (int i = threadIdx.x; i < T; i += 256) {
      tokens[(int64_t)b * T + i] = s_tok[i];
      scores[(int64_t)b * T + i] = s_sc[i];
    }
  }
}

static inline int ew_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace dn

using namespace dn;

extern "C" int dn_rmsnorm(const float* x, int32_t ldx, void* y, int32_t ldy, int32_t out_dtype, int32_t M, int32_t D, int32_t T,
                          const float* gamma, const float* gamma_beta, int32_t gb_ld, int32_t gb_half, void* stream) {
  DN_CHECK_ARG(x && y && M > 0 && D > 0 && T > 0, "dn_rmsnorm: bad args");
  DN_CHECK_ARG(D % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldy >= D && ldx >= D, "dn_rmsnorm: D=%d ldx=%d ldy=%d must be multiples of 4", D, ldx, ldy);
  DN_CHECK_ARG(!gamma_beta || (gb_ld % 4 == 0 && gb_half % 4 == 0), "dn_rmsnorm: gamma_beta strides must be multiples of 4");
  hipLaunchKernelGGL(rmsnorm_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, out_dtype, M, D, T, gamma,
                     gamma_beta, gb_ld, gb_half);
  DN_CHECK_LAUNCH("dn_rmsnorm");
  return DN_OK;
}

extern "C" int dn_time_cond(const int32_t* times, int32_t B, const float* w_freq, int32_t half, const float* W, const float* bias,
                            int32_t C, float* out, void* out_act, int32_t act_dtype, int32_t ldo, void* stream) {
  DN_CHECK_ARG(times && w_freq && W && bias && out && B > 0 && half > 0 && C > 0 && ldo >= C, "dn_time_cond: bad args");
  hipLaunchKernelGGL(time_cond_kernel, dim3((C + 15) / 16, B), dim3(256), (2 * half + 1) * sizeof(float), (hipStream_t)stream, times,
                     w_freq, half, W, bias, C, out, out_act, act_dtype, ldo);
  DN_CHECK_LAUNCH("dn_time_cond");
  return DN_OK;
}

extern "C" int dn_ddim_step(const float* x, const float* eps, float* x_out, void* x_act, int32_t act_dtype, int32_t ld_act, int32_t M,
                            int32_t C, int32_t ld, int32_t T, const float* coef, const int32_t* t, void* stream) {
  DN_CHECK_ARG(x && eps && x_out && coef && t && M > 0 && C > 0 && ld >= C && T > 0, "dn_ddim_step: bad args");
  hipLaunchKernelGGL(ddim_step_kernel, dim3(ew_grid((int64_t)M * C)), dim3(256), 0, (hipStream_t)stream, x, eps, x_out, x_act, act_dtype,
                     ld_act, M, C, ld, T, coef, t);
  DN_CHECK_LAUNCH("dn_ddim_step");
  return DN_OK;
}

// forward_with_cond_scale's combination (latent_module.py:813-826): out = null + (cond - null) * scale over the two halves of one
// 2B-batch pass ([cond rows ; null rows])
__global__ __launch_bounds__(256) void cfg_combine_kernel(const float* __restrict__ both, float scale, int64_t n, float* __restrict__ out) {
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float c = both[i], u = both[n + i];
    out[i] = __fadd_rn(u, __fmul_rn(__fsub_rn(c, u), scale));
  }
}
extern "C" int dn_cfg_combine(const float* both, float scale, int64_t n, float* out, void* stream) {
  DN_CHECK_ARG(both && out && n > 0, "dn_cfg_combine: bad argument");
  hipLaunchKernelGGL(cfg_combine_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, both, scale, n, out);
  DN_CHECK_LAUNCH("dn_cfg_combine");
  return DN_OK;
}

// (engine.hip: the ancestral update of dn_ddpm_loop)
int dn_ddpm_step_launch(float* x, const float* eps, int M, int C, int T, const float* table, const int32_t* t, int clip, const float* noise,
                        int64_t noise_row, int t_top, uint64_t seed, hipStream_t stream) {
  DN_CHECK_ARG(C % 4 == 0, "dn_ddpm_loop: the latent width must be a multiple of 4");
  hipLaunchKernelGGL(ddpm_step_kernel, dim3(ew_grid(((int64_t)M * C) >> 2)), dim3(256), 0, stream, x, eps, M, C, T, table, t, clip, noise,
                     noise_row, t_top, seed);
  DN_CHECK_LAUNCH("dn_ddpm_loop step");
  return DN_OK;
}

extern "C" int dn_q_sample(const float* x, const float* noise, float* out, void* out_act, int32_t act_dtype, int32_t ld_act, int32_t M,
                           int32_t C, int32_t ld, int32_t T, const float* coef_a, const float* coef_b, const int32_t* t, void* stream) {
  DN_CHECK_ARG(x && noise && out && coef_a && coef_b && t && M > 0 && C > 0 && ld >= C && T > 0, "dn_q_sample: bad args");
  hipLaunchKernelGGL(q_sample_kernel, dim3(ew_grid((int64_t)M * C)), dim3(256), 0, (hipStream_t)stream, x, noise, out, out_act, act_dtype,
                     ld_act, M, C, ld, T, coef_a, coef_b, t);
  DN_CHECK_LAUNCH("dn_q_sample");
  return DN_OK;
}

extern "C" int dn_posterior_sample(const float* params, int32_t ldp, const float* noise, int32_t ldn, float* z, void* z_act,
                                   int32_t act_dtype, int32_t ldz, int32_t M, int32_t Z, int32_t T, const int32_t* lengths,
                                   float* kl_rows, void* stream) {
  DN_CHECK_ARG(params && noise && z && M > 0 && Z > 0 && ldp >= 2 * Z && ldn >= Z && ldz >= Z && T > 0, "dn_posterior_sample: bad args");
  hipLaunchKernelGGL(posterior_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, params, ldp, noise, ldn, z, z_act, act_dtype,
                     ldz, M, Z, T, lengths, kl_rows);
  DN_CHECK_LAUNCH("dn_posterior_sample");
  return DN_OK;
}

extern "C" int dn_argmax_units(const float* logits, int32_t ld, int32_t M, int32_t V, int32_t offset, int32_t* units, void* stream) {
  DN_CHECK_ARG(logits && units && M > 0 && V > 0 && ld >= V, "dn_argmax_units: bad args");
  hipLaunchKernelGGL(argmax_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld, M, V, offset, units);
  DN_CHECK_LAUNCH("dn_argmax_units");
  return DN_OK;
}

extern "C" int dn_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
  DN_CHECK_ARG(out && n > 0, "dn_randn: bad args");
  hipLaunchKernelGGL(randn_kernel, dim3(ew_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, out, n, seed, offset);
  DN_CHECK_LAUNCH("dn_randn");
  return DN_OK;
}

extern "C" int dn_convert_rows(const void* src, int32_t src_dtype, int32_t lds, void* dst, int32_t dst_dtype, int32_t ldd, int32_t M,
                               int32_t C, void* stream) {
  DN_CHECK_ARG(src && dst && M > 0 && C > 0 && lds >= C && ldd >= C, "dn_convert_rows: bad args");
  hipLaunchKernelGGL(convert_rows_kernel, dim3(ew_grid((int64_t)M * ldd)), dim3(256), 0, (hipStream_t)stream, src, src_dtype, lds, dst,
                     dst_dtype, ldd, M, C);
  DN_CHECK_LAUNCH("dn_convert_rows");
  return DN_OK;
}

extern "C" int dn_gaussian_step(const DnGaussianStep* p, void* stream) {
  DN_CHECK_ARG(p && p->x && p->model_out && p->sample && p->t && p->table, "dn_gaussian_step: null argument");
  DN_CHECK_ARG(p->N > 0 && p->inner > 0 && (p->sampler == 0 || p->sampler == 1), "dn_gaussian_step: bad shape / sampler");
  hipLaunchKernelGGL(gaussian_step_kernel, dim3(ew_grid((int64_t)p->N * p->inner)), dim3(256), 0, (hipStream_t)stream, *p);
  DN_CHECK_LAUNCH("dn_gaussian_step");
  return DN_OK;
}

extern "C" int dn_gaussian_moments(const DnGaussianMoments* p, void* stream) {
  DN_CHECK_ARG(p && p->x && p->t && p->table && (p->model_out || p->x_start), "dn_gaussian_moments: null argument");
  DN_CHECK_ARG(p->N > 0 && p->inner > 0, "dn_gaussian_moments: bad shape");
  hipLaunchKernelGGL(gaussian_moments_kernel, dim3(ew_grid((int64_t)p->N * p->inner)), dim3(256), 0, (hipStream_t)stream, *p);
  DN_CHECK_LAUNCH("dn_gaussian_moments");
  return DN_OK;
}

extern "C" int dn_cmlm_step(const float* logits, int32_t* tokens, float* scores, int32_t* predicted, int32_t B, int32_t T, int32_t V,
                            int32_t step, int32_t max_step, int32_t unk, int32_t pad, void* stream) {
  DN_CHECK_ARG(logits && tokens && scores && predicted && B > 0 && T > 0 && T <= 2048 && V > 1 && max_step > 0 && step >= 0,
               "dn_cmlm_step: bad args (T=%d must be <= 2048)", T);
  const int remask = (step + 1) < max_step;
  const float p = (float)(1.0 - (double)(step + 1) / (double)max_step);  // the Python float of the reference, cast to the scores' fp32
  hipLaunchKernelGGL(cmlm_step_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, tokens, scores, predicted, T, V, p, remask, unk, pad,
                     (const int32_t*)nullptr, max_step);
  DN_CHECK_LAUNCH("dn_cmlm_step");
  return DN_OK;
}

extern "C" int dn_cmlm_step_dev(const float* logits, int32_t* tokens, float* scores, int32_t* predicted, int32_t B, int32_t T, int32_t V,
                                const int32_t* step_dev, int32_t max_step, int32_t unk, int32_t pad, void* stream) {
  DN_CHECK_ARG(logits && tokens && scores && predicted && step_dev && B > 0 && T > 0 && T <= 2048 && V > 1 && max_step > 0,
               "dn_cmlm_step_dev: bad args (T=%d must be <= 2048)", T);
  hipLaunchKernelGGL(cmlm_step_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, tokens, scores, predicted, T, V, 0.f, 0, unk, pad,
                     step_dev, max_step);
  DN_CHECK_LAUNCH("dn_cmlm_step_dev");
  return DN_OK;
}
