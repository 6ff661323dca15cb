// dn_attention_backward: gradient of the fused key-masked multi-head self-attention (reference Attend.forward non-flash
// branch, latent_module.py:299-343; softmax(Q K^T scale) V with keys >= length masked), flash-style: P is recomputed from
// the forward's per-query log-sum-exp, no [B,H,T,T] tensor exists in either direction.
//
//   delta[q] = sum_d dO[q,d] O[q,d]        P = 2^(S sc - lse[q])        dS = P o (dO V^T - delta[q])
//   dV = P^T dO          dK = dS^T Q * scale          dQ = dS K * scale
//
// Two kernels, each the forward kernel's transposed-product scheme (attention.hip) with the roles permuted, so that the
// second product of each takes its B operand straight from the first product's accumulators (no LDS round trip for P/dS)
// and there are no atomics (dQ and dK/dV each have exactly one writer: bit-reproducible):
//   dkv kernel: a wave owns 32 keys (K, V fragments in registers), walks the queries in tiles of 64 staged in LDS:
//       S = Q K^T (A = Q rows, B = K frags) and dP = dO V^T -> P, dS with the QUERY on the accumulator row;
//       dV^T += dO^T P, dK^T += Q^T dS: A = the transposed-read (ds_read_b64_tr_b16 / scalar f32) Q and dO tiles.
//   dq kernel: a wave owns 32 queries (Q, dO fragments in registers), walks the keys in tiles of 64 (exactly the forward):
//       S^T = K Q^T, dP^T = V dO^T -> dS^T with the KEY on the accumulator row;  dQ^T += K^T dS^T.
// Output lanes hold 4 consecutive head dims of one row, as the forward's O store.
#include <initializer_list>
#include <type_traits>

#include "common.h"

namespace dn {

template <typename E> struct BwdGeom;
template <> struct BwdGeom<BF16> { static constexpr int ROWB = 256; };
template <> struct BwdGeom<F32> { static constexpr int ROWB = 512; };

template <typename E>
__device__ __forceinline__ int bwd_lds_off(int row, int byte);
template <>
__device__ __forceinline__ int bwd_lds_off<BF16>(int row, int byte) {
  return row * 256 + ((((byte >> 5) ^ (row & 7))) << 5) + (byte & 31);
}
template <>
__device__ __forceinline__ int bwd_lds_off<F32>(int row, int byte) {
  return row * 512 + (((byte >> 4) ^ (row & 15)) << 4) + (byte & 15);
}

constexpr int BT = 64;  // rows of a staged tile (keys in the dq kernel, queries in the dkv kernel)
constexpr float LSE_BIG = 3.0e38f;

// delta[b, h, q] = sum_d dO[q, h*dh + d] * O[q, h*dh + d].  G lanes (a power of two, 8 elements each) per (row, head), 64 / G
// consecutive heads per wave: a wave reads whole contiguous pieces of a row of both tensors.
__global__ __launch_bounds__(256) void attn_delta_kernel(const void* __restrict__ o, int ldo, const void* __restrict__ dout, int lddo, int dtype,
                                                         int B, int T, int heads, int dh, int G, float* __restrict__ delta) {
  const int lane = threadIdx.x & 63, per = 64 / G;
  const int sub = lane & (G - 1);
  const int64_t idx = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * per + lane / G;
  const bool live = idx < (int64_t)B * T * heads;
  const int h = live ? (int)(idx % heads) : 0;
  const int64_t m = live ? idx / heads : 0;
  float s = 0.f;
  if (live)
    for (int d = sub * 8; d < dh; d += G * 8) {
      const float4 a = load4(o, m * ldo + h * dh + d, dtype), g = load4(dout, m * lddo + h * dh + d, dtype);
      s += a.x * g.x + a.y * g.y + a.z * g.z + a.w * g.w;
      if (d + 4 < dh) {
        const float4 a2 = load4(o, m * ldo + h * dh + d + 4, dtype), g2 = load4(dout, m * lddo + h * dh + d + 4, dtype);
        s += a2.x * g2.x + a2.y * g2.y + a2.z * g2.z + a2.w * g2.w;
      }
    }
  for (int off = G >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (live && sub == 0) {
    const int b = (int)(m / T), q = (int)(m - (int64_t)b * T);
    delta[((int64_t)b * heads + h) * T + q] = s;
  }
}

// The tile machinery shared by both kernels: two row-major [BT][DHP] tiles (X, Y) of one (batch, head), double-buffered in
// LDS, one register set of lead (tile t+1 is requested at the top of tile t and written to the other buffer at its end).
template <typename E, int DHP, int NT = 256>
struct TilePair {
  static constexpr int ES = Elem<E>::bytes;
  static constexpr int ROWB = BwdGeom<E>::ROWB;
  static constexpr int NCH = DHP * ES / 16;
  static constexpr int TILE_LDS = BT * ROWB;
  static constexpr int NPT = (BT * NCH + NT - 1) / NT;
  uint4 xr[NPT], yr[NPT];
  __device__ __forceinline__ void issue(const char* xp, int64_t ldxb, const char* yp, int64_t ldyb, int row0, int T, int dhb, int tid) {
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int idx = tid + i * NT;
      const int row = idx / NCH, ch = idx - row * NCH;
      const int r = row0 + row;
      xr[i] = yr[i] = make_uint4(0, 0, 0, 0);
      if (idx < BT * NCH && r < T && ch * 16 < dhb) {
        xr[i] = *reinterpret_cast<const uint4*>(xp + (int64_t)r * ldxb + ch * 16);
        yr[i] = *reinterpret_cast<const uint4*>(yp + (int64_t)r * ldyb + ch * 16);
      }
    }
  }
  __device__ __forceinline__ void write(char* x_lds, char* y_lds, int buf, int tid) {
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int idx = tid + i * NT;
      const int row = idx / NCH, ch = idx - row * NCH;
      if (idx < BT * NCH) {
        const int off = buf * TILE_LDS + bwd_lds_off<E>(row, ch * 16);
        *reinterpret_cast<uint4*>(x_lds + off) = xr[i];
        *reinterpret_cast<uint4*>(y_lds + off) = yr[i];
      }
    }
  }
};

// A-operand fragment of a transposed product from a row-major tile: the k-slots of 16-row tiles (2kk, 2kk+1) of the tile,
// output row = column dt*16 + fr of the tile (bf16: two transposed 8-byte reads).
template <typename E>
__device__ __forceinline__ uint4 tr_frag(const char* tile, int base, int kk, int rowb);
template <>
__device__ __forceinline__ uint4 tr_frag<BF16>(const char* tile, int base, int kk, int rowb) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + base + (2 * kk) * 16 * rowb));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + base + (2 * kk + 1) * 16 * rowb));
  const uint2 lo2 = __builtin_bit_cast(uint2, lo), hi2 = __builtin_bit_cast(uint2, hi);
  return make_uint4(lo2.x, lo2.y, hi2.x, hi2.y);
}

struct AttnBwdArgs {
  const void *q, *k, *v, *dout;
  void *dq, *dk, *dv;
  int ldq, ldk, ldv, lddo, lddq, lddk, lddv;
  int B, T, heads, dim_head, dtype;
  const int32_t* lengths;
  float scale;
  const float* lse;
  const float* delta;
  float dropout_p;  // the forward's attention dropout: out = (P o M / (1 - p)) V, so dV uses the dropped P and dP = (dO V^T) o M / (1 - p)
  uint32_t seed_lo, seed_hi;
};

// ------------------------------------------------------------------------------------------------------ dK, dV
// WT: 16-key tiles per wave.  2 = four waves of 32 keys; 1 = eight waves of 16 keys (the same 128 keys and LDS per workgroup, half the
// K / V fragments and accumulators per wave: twice the waves per SIMD -- see attn_kernel's QT; option attn_waves8).
template <typename E, int DHP, bool DROP, int WT = 2>
__global__ __launch_bounds__(128 / (16 * WT) * 64, WT == 1 ? (DHP <= 64 ? 4 : 2) : 1) void attn_bwd_dkv_kernel(const AttnBwdArgs p) {
  constexpr int NT = 128 / (16 * WT) * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using TP = TilePair<E, DHP, NT>;
  constexpr int ES = Elem<E>::bytes;
  constexpr int ROWB = BwdGeom<E>::ROWB;
  constexpr int KS_D = DHP * ES / 64;
  constexpr int DT = DHP / 16;
  constexpr int TILE_LDS = TP::TILE_LDS;
  char* q_lds = smem;                     // [2][BT][ROWB]
  char* do_lds = smem + 2 * TILE_LDS;     // [2][BT][ROWB]
  float* st_lds = reinterpret_cast<float*>(smem + 4 * TILE_LDS);  // [2][2][BT]: lse, delta of the tile's queries

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  int rblk, h, b;  // this workgroup's row block of (batch b, head h), XCD-aware order
  dn_xcd_block_map(rblk, h, b);
  const int T = p.T, dh = p.dim_head;
  const int k0 = rblk * 128 + wave * (16 * WT);  // this wave's keys
  const int dhb = dh * ES;

  const char* qp = reinterpret_cast<const char*>(p.q) + ((int64_t)b * T * p.ldq + h * dh) * ES;
  const char* kp = reinterpret_cast<const char*>(p.k) + ((int64_t)b * T * p.ldk + h * dh) * ES;
  const char* vp = reinterpret_cast<const char*>(p.v) + ((int64_t)b * T * p.ldv + h * dh) * ES;
  const char* dop = reinterpret_cast<const char*>(p.dout) + ((int64_t)b * T * p.lddo + h * dh) * ES;
  const float* lsep = p.lse + ((int64_t)b * p.heads + h) * T;
  const float* delp = p.delta + ((int64_t)b * p.heads + h) * T;

  int len = p.lengths ? p.lengths[b] : T;
  len = len < T ? len : T;
  float sc = p.scale * 1.44269504088896340736f;
  float out_scale = p.scale;
  if (len <= 0) {  // every key masked: uniform softmax, scores constant -> no gradient reaches q / k
    len = T;
    sc = 0.f;
    out_scale = 0.f;
  }

  uint32_t drop_thr = 0;
  float drop_inv = 1.f;
  if constexpr (DROP) {
    drop_thr = (uint32_t)fminf(p.dropout_p * 4294967296.0f, 4294967040.0f);
    drop_inv = 1.0f / (1.0f - p.dropout_p);
  }
  // K, V fragments (B operands): row = key
  uint4 kf[WT][KS_D], vf[WT][KS_D];
#pragma unroll
  for (int kt = 0; kt < WT; ++kt) {
    int key = k0 + kt * 16 + fr;
    key = key < T ? key : T - 1;
#pragma unroll
    for (int ks = 0; ks < KS_D; ++ks) {
      const int byte = ks * 64 + fg * 16;
      const bool ok = byte < dhb;
      kf[kt][ks] = ok ? *reinterpret_cast<const uint4*>(kp + (int64_t)key * p.ldk * ES + byte) : make_uint4(0, 0, 0, 0);
      vf[kt][ks] = ok ? *reinterpret_cast<const uint4*>(vp + (int64_t)key * p.ldv * ES + byte) : make_uint4(0, 0, 0, 0);
    }
  }
  f32x4 acc_dk[DT][WT], acc_dv[DT][WT];  // [d tile][key tile]: lane holds d = dt*16 + 4 fg + r, key = kt*16 + fr
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int kt = 0; kt < WT; ++kt) acc_dk[i][kt] = acc_dv[i][kt] = f32x4{0.f, 0.f, 0.f, 0.f};

  int a_base[KS_D];  // A rows of the first products: row fr of a 16-query tile, chunk ks*4 + fg
#pragma unroll
  for (int ks = 0; ks < KS_D; ++ks) a_base[ks] = bwd_lds_off<E>(fr, ks * 64 + fg * 16);
  int t_base[DT];    // transposed reads (bf16): row fg*4 + (fr >> 2), 8-byte piece dt*4 + (fr & 3)
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) t_base[dt] = bwd_lds_off<E>(fg * 4 + (fr >> 2), ES == 2 ? dt * 32 + (fr & 3) * 8 : 0);

  TP tp;
  auto stage_stats = [&](int buf, int q0) {  // lse (+BIG past the end, so P = 0 there) and delta of queries q0 .. q0+63
    if (tid < 2 * BT) {
      const int which = tid / BT, i = tid - which * BT;
      const int q = q0 + i;
      float v = which == 0 ? LSE_BIG : 0.f;
      if (q < T) v = which == 0 ? lsep[q] : delp[q];
      st_lds[(buf * 2 + which) * BT + i] = v;
    }
  };
  tp.issue(qp, (int64_t)p.ldq * ES, dop, (int64_t)p.lddo * ES, 0, T, dhb, tid);
  tp.write(q_lds, do_lds, 0, tid);
  stage_stats(0, 0);
  __syncthreads();

  int buf = 0;
  for (int q0 = 0; q0 < T; q0 += BT, buf ^= 1) {
    const bool more = q0 + BT < T;
    if (more) tp.issue(qp, (int64_t)p.ldq * ES, dop, (int64_t)p.lddo * ES, q0 + BT, T, dhb, tid);
    const char* qt_lds = q_lds + buf * TILE_LDS;
    const char* dot_lds = do_lds + buf * TILE_LDS;
    const float* lse_t = st_lds + (buf * 2 + 0) * BT;
    const float* del_t = st_lds + (buf * 2 + 1) * BT;

    // ---- S = Q K^T, dP = dO V^T : acc[qt][kt][r] <-> query qt*16 + 4 fg + r, key kt*16 + fr
    f32x4 acc_s[4][WT], acc_dp[4][WT];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
#pragma unroll
      for (int kt = 0; kt < WT; ++kt) acc_s[qt][kt] = acc_dp[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS_D; ++ks) {
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        const uint4 qa = *reinterpret_cast<const uint4*>(qt_lds + a_base[ks] + qt * 16 * ROWB);
        const uint4 da = *reinterpret_cast<const uint4*>(dot_lds + a_base[ks] + qt * 16 * ROWB);
#pragma unroll
        for (int kt = 0; kt < WT; ++kt) {
          mma_kstep<E>(acc_s[qt][kt], qa, kf[kt][ks]);
          mma_kstep<E>(acc_dp[qt][kt], da, vf[kt][ks]);
        }
      }
    }
    // ---- P = 2^(S sc - lse[q]) (0 for masked keys), dS = P (dP - delta[q]); acc_s <- P, acc_dp <- dS
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      const float4 l4 = *reinterpret_cast<const float4*>(lse_t + qt * 16 + fg * 4);
      const float4 d4 = *reinterpret_cast<const float4*>(del_t + qt * 16 + fg * 4);
      const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq_[4] = {d4.x, d4.y, d4.z, d4.w};
      uint32_t rowh[4] = {0, 0, 0, 0};
      if constexpr (DROP) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          rowh[r] = dn_drop_row((uint32_t)(((int64_t)b * p.heads + h) * T + (q0 + qt * 16 + fg * 4 + r)), p.seed_lo);
      }
#pragma unroll
      for (int kt = 0; kt < WT; ++kt) {
        const int key = k0 + kt * 16 + fr;
        const bool key_ok = key < len;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pv = __builtin_amdgcn_exp2f(fmaf(acc_s[qt][kt][r], sc, -lq[r]));
          pv = key_ok ? pv : 0.f;
          float md = 1.f;  // dropout mask / (1 - p)
          if constexpr (DROP) md = dn_drop_keep(rowh[r], (uint32_t)key, p.seed_hi, drop_thr) ? drop_inv : 0.f;
          acc_s[qt][kt][r] = pv * md;                                      // dropped P: dV = (P o M)^T dO
          acc_dp[qt][kt][r] = pv * (acc_dp[qt][kt][r] * md - dq_[r]);      // dS = P (dP o M - delta)
        }
      }
    }
    // ---- dV^T += dO^T P ; dK^T += Q^T dS   (contraction over the tile's 64 queries)
    if constexpr (ES == 2) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {  // 32-query k-steps = accumulator tiles (2kk, 2kk+1)
        uint4 pf[WT], sf[WT];
#pragma unroll
        for (int kt = 0; kt < WT; ++kt) {
          pf[kt].x = pack_bf16x2(acc_s[2 * kk][kt][0], acc_s[2 * kk][kt][1]);
          pf[kt].y = pack_bf16x2(acc_s[2 * kk][kt][2], acc_s[2 * kk][kt][3]);
          pf[kt].z = pack_bf16x2(acc_s[2 * kk + 1][kt][0], acc_s[2 * kk + 1][kt][1]);
          pf[kt].w = pack_bf16x2(acc_s[2 * kk + 1][kt][2], acc_s[2 * kk + 1][kt][3]);
          sf[kt].x = pack_bf16x2(acc_dp[2 * kk][kt][0], acc_dp[2 * kk][kt][1]);
          sf[kt].y = pack_bf16x2(acc_dp[2 * kk][kt][2], acc_dp[2 * kk][kt][3]);
          sf[kt].z = pack_bf16x2(acc_dp[2 * kk + 1][kt][0], acc_dp[2 * kk + 1][kt][1]);
          sf[kt].w = pack_bf16x2(acc_dp[2 * kk + 1][kt][2], acc_dp[2 * kk + 1][kt][3]);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const uint4 dof = tr_frag<BF16>(dot_lds, t_base[dt], kk, ROWB);
          const uint4 qtf = tr_frag<BF16>(qt_lds, t_base[dt], kk, ROWB);
#pragma unroll
          for (int kt = 0; kt < WT; ++kt) {
            mma_kstep<E>(acc_dv[dt][kt], dof, pf[kt]);
            mma_kstep<E>(acc_dk[dt][kt], qtf, sf[kt]);
          }
        }
      }
    } else {
#pragma unroll
      for (int qt = 0; qt < 4; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qrow = qt * 16 + fg * 4 + r;  // k-slot fg of MFMA (qt, r)
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const float dov = *reinterpret_cast<const float*>(dot_lds + bwd_lds_off<E>(qrow, (dt * 16 + fr) * 4));
            const float qv = *reinterpret_cast<const float*>(qt_lds + bwd_lds_off<E>(qrow, (dt * 16 + fr) * 4));
#pragma unroll
            for (int kt = 0; kt < WT; ++kt) {
              acc_dv[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dov, acc_s[qt][kt][r], acc_dv[dt][kt], 0, 0, 0);
              acc_dk[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv, acc_dp[qt][kt][r], acc_dk[dt][kt], 0, 0, 0);
            }
          }
        }
    }
    if (more) {
      tp.write(q_lds, do_lds, buf ^ 1, tid);
      stage_stats(buf ^ 1, q0 + BT);
    }
    __syncthreads();
  }

  char* dkp = reinterpret_cast<char*>(p.dk);
  char* dvp = reinterpret_cast<char*>(p.dv);
#pragma unroll
  for (int kt = 0; kt < WT; ++kt) {
    const int key = k0 + kt * 16 + fr;
    if (key >= T) continue;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + fg * 4;
      if (d >= dh) continue;
      const f32x4 gk = acc_dk[dt][kt], gv = acc_dv[dt][kt];
      store4(dkp, ((int64_t)b * T + key) * p.lddk + h * dh + d, p.dtype, gk[0] * out_scale, gk[1] * out_scale, gk[2] * out_scale,
             gk[3] * out_scale);
      store4(dvp, ((int64_t)b * T + key) * p.lddv + h * dh + d, p.dtype, gv[0], gv[1], gv[2], gv[3]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------ dQ
// WT: 16-query tiles per wave (2 = four waves of 32 queries, 1 = eight waves of 16: see the dK/dV kernel).
// (WT = 2: two waves per SIMD asked for where the kernel fits 256 registers without spilling: left alone the compiler spent 260 on the
// 96-dim head -- the VAE's -- which made that variant a one-wave-per-SIMD kernel; 236 when told)
template <typename E, int DHP, bool DROP, int WT = 2>
__global__ __launch_bounds__(128 / (16 * WT) * 64, WT == 1 ? (DHP <= 64 ? 4 : 2) : (std::is_same<E, BF16>::value && DHP <= 96) ? 2 : 1) void attn_bwd_dq_kernel(const AttnBwdArgs p) {
  constexpr int NT = 128 / (16 * WT) * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using TP = TilePair<E, DHP, NT>;
  constexpr int ES = Elem<E>::bytes;
  constexpr int ROWB = BwdGeom<E>::ROWB;
  constexpr int KS_D = DHP * ES / 64;
  constexpr int DT = DHP / 16;
  constexpr int TILE_LDS = TP::TILE_LDS;
  char* k_lds = smem;
  char* v_lds = smem + 2 * TILE_LDS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  int rblk, h, b;  // this workgroup's row block of (batch b, head h), XCD-aware order
  dn_xcd_block_map(rblk, h, b);
  const int T = p.T, dh = p.dim_head;
  const int q0 = rblk * 128 + wave * (16 * WT);
  const int dhb = dh * ES;

  const char* qp = reinterpret_cast<const char*>(p.q) + ((int64_t)b * T * p.ldq + h * dh) * ES;
  const char* kp = reinterpret_cast<const char*>(p.k) + ((int64_t)b * T * p.ldk + h * dh) * ES;
  const char* vp = reinterpret_cast<const char*>(p.v) + ((int64_t)b * T * p.ldv + h * dh) * ES;
  const char* dop = reinterpret_cast<const char*>(p.dout) + ((int64_t)b * T * p.lddo + h * dh) * ES;
  const float* lsep = p.lse + ((int64_t)b * p.heads + h) * T;
  const float* delp = p.delta + ((int64_t)b * p.heads + h) * T;

  int len = p.lengths ? p.lengths[b] : T;
  len = len < T ? len : T;
  float sc = p.scale * 1.44269504088896340736f;
  float out_scale = p.scale;
  if (len <= 0) {
    len = T;
    sc = 0.f;
    out_scale = 0.f;
  }

  uint4 qf[WT][KS_D], dof[WT][KS_D];  // B operands: row = query
  float lse_q[WT], del_q[WT];
  uint32_t drop_thr = 0, rowh[WT];
#pragma unroll
  for (int qt = 0; qt < WT; ++qt) rowh[qt] = 0;
  float drop_inv = 1.f;
  if constexpr (DROP) {
    drop_thr = (uint32_t)fminf(p.dropout_p * 4294967296.0f, 4294967040.0f);
    drop_inv = 1.0f / (1.0f - p.dropout_p);
  }
#pragma unroll
  for (int qt = 0; qt < WT; ++qt) {
    int q = q0 + qt * 16 + fr;
    if constexpr (DROP) rowh[qt] = dn_drop_row((uint32_t)(((int64_t)b * p.heads + h) * T + q), p.seed_lo);
    q = q < T ? q : T - 1;
    lse_q[qt] = lsep[q];
    del_q[qt] = delp[q];
#pragma unroll
    for (int ks = 0; ks < KS_D; ++ks) {
      const int byte = ks * 64 + fg * 16;
      const bool ok = byte < dhb;
      qf[qt][ks] = ok ? *reinterpret_cast<const uint4*>(qp + (int64_t)q * p.ldq * ES + byte) : make_uint4(0, 0, 0, 0);
      dof[qt][ks] = ok ? *reinterpret_cast<const uint4*>(dop + (int64_t)q * p.lddo * ES + byte) : make_uint4(0, 0, 0, 0);
    }
  }
  f32x4 acc_dq[DT][WT];  // lane holds d = dt*16 + 4 fg + r, query qt*16 + fr
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int qt = 0; qt < WT; ++qt) acc_dq[i][qt] = f32x4{0.f, 0.f, 0.f, 0.f};

  int a_base[KS_D];
#pragma unroll
  for (int ks = 0; ks < KS_D; ++ks) a_base[ks] = bwd_lds_off<E>(fr, ks * 64 + fg * 16);
  int t_base[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) t_base[dt] = bwd_lds_off<E>(fg * 4 + (fr >> 2), ES == 2 ? dt * 32 + (fr & 3) * 8 : 0);

  TP tp;
  tp.issue(kp, (int64_t)p.ldk * ES, vp, (int64_t)p.ldv * ES, 0, T, dhb, tid);
  tp.write(k_lds, v_lds, 0, tid);
  __syncthreads();

  int buf = 0;
  for (int kv0 = 0; kv0 < len; kv0 += BT, buf ^= 1) {
    const bool more = kv0 + BT < len;
    if (more) tp.issue(kp, (int64_t)p.ldk * ES, vp, (int64_t)p.ldv * ES, kv0 + BT, T, dhb, tid);
    const char* kt_lds = k_lds + buf * TILE_LDS;
    const char* vt_lds = v_lds + buf * TILE_LDS;

    // ---- S^T = K Q^T, dP^T = V dO^T : acc[kt][qt][r] <-> key kt*16 + 4 fg + r, query qt*16 + fr
    f32x4 acc_s[4][WT], acc_dp[4][WT];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int qt = 0; qt < WT; ++qt) acc_s[kt][qt] = acc_dp[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS_D; ++ks) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const uint4 ka = *reinterpret_cast<const uint4*>(kt_lds + a_base[ks] + kt * 16 * ROWB);
        const uint4 va = *reinterpret_cast<const uint4*>(vt_lds + a_base[ks] + kt * 16 * ROWB);
#pragma unroll
        for (int qt = 0; qt < WT; ++qt) {
          mma_kstep<E>(acc_s[kt][qt], ka, qf[qt][ks]);
          mma_kstep<E>(acc_dp[kt][qt], va, dof[qt][ks]);
        }
      }
    }
    // ---- dS^T = P^T (dP^T - delta[q]) -> acc_dp
    const bool straddle = kv0 + BT > len;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int qt = 0; qt < WT; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kv0 + kt * 16 + fg * 4 + r;
          float pv = __builtin_amdgcn_exp2f(fmaf(acc_s[kt][qt][r], sc, -lse_q[qt]));
          if (straddle) pv = key < len ? pv : 0.f;
          float md = 1.f;
          if constexpr (DROP) md = dn_drop_keep(rowh[qt], (uint32_t)key, p.seed_hi, drop_thr) ? drop_inv : 0.f;
          acc_dp[kt][qt][r] = pv * (acc_dp[kt][qt][r] * md - del_q[qt]);
        }
    // ---- dQ^T += K^T dS^T  (contraction over the tile's 64 keys)
    if constexpr (ES == 2) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        uint4 sf[WT];
#pragma unroll
        for (int qt = 0; qt < WT; ++qt) {
          sf[qt].x = pack_bf16x2(acc_dp[2 * kk][qt][0], acc_dp[2 * kk][qt][1]);
          sf[qt].y = pack_bf16x2(acc_dp[2 * kk][qt][2], acc_dp[2 * kk][qt][3]);
          sf[qt].z = pack_bf16x2(acc_dp[2 * kk + 1][qt][0], acc_dp[2 * kk + 1][qt][1]);
          sf[qt].w = pack_bf16x2(acc_dp[2 * kk + 1][qt][2], acc_dp[2 * kk + 1][qt][3]);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const uint4 ktf = tr_frag<BF16>(kt_lds, t_base[dt], kk, ROWB);
#pragma unroll
          for (int qt = 0; qt < WT; ++qt) mma_kstep<E>(acc_dq[dt][qt], ktf, sf[qt]);
        }
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int krow = kt * 16 + fg * 4 + r;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const float kv = *reinterpret_cast<const float*>(kt_lds + bwd_lds_off<E>(krow, (dt * 16 + fr) * 4));
#pragma unroll
            for (int qt = 0; qt < WT; ++qt) acc_dq[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv, acc_dp[kt][qt][r], acc_dq[dt][qt], 0, 0, 0);
          }
        }
    }
    if (more) tp.write(k_lds, v_lds, buf ^ 1, tid);
    __syncthreads();
  }

  char* dqp = reinterpret_cast<char*>(p.dq);
#pragma unroll
  for (int qt = 0; qt < WT; ++qt) {
    const int q = q0 + qt * 16 + fr;
    if (q >= T) continue;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + fg * 4;
      if (d >= dh) continue;
      const f32x4 g = acc_dq[dt][qt];
      store4(dqp, ((int64_t)b * T + q) * p.lddq + h * dh + d, p.dtype, g[0] * out_scale, g[1] * out_scale, g[2] * out_scale,
             g[3] * out_scale);
    }
  }
}

// WK / WQ: tiles per wave of the dK/dV and of the dQ kernel
template <typename E, int DHP, bool DROP, int WK = 2, int WQ = 2>
static int launch_attn_bwd_v(const AttnBwdArgs& a, hipStream_t s) {
  constexpr int tile = BT * BwdGeom<E>::ROWB;
  constexpr int lds_dkv = 4 * tile + 4 * BT * (int)sizeof(float);
  constexpr int lds_dq = 4 * tile;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<E, DHP, DROP, WK>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_dkv);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<E, DHP, DROP, WQ>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_dq);
    attr_done = true;
  }
  dim3 grid((a.T + 127) / 128, a.heads, a.B);
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<E, DHP, DROP, WK>), grid, dim3(128 / (16 * WK) * 64), lds_dkv, s, a);
  hipLaunchKernelGGL((attn_bwd_dq_kernel<E, DHP, DROP, WQ>), grid, dim3(128 / (16 * WQ) * 64), lds_dq, s, a);
  DN_CHECK_LAUNCH("dn_attention_backward");
  return DN_OK;
}

template <typename E, int DHP>
static int launch_attn_bwd(const AttnBwdArgs& a, hipStream_t s) {
  // option attn_waves8 (default on), 2-byte mode: eight waves of one tile per workgroup where that raises the waves per SIMD -- the
  // dK/dV kernel at every head size (96 dims: 310 -> 214 registers, one -> two waves per SIMD), the dQ kernel up to 64 dims (above,
  // its 32-query form already holds two waves per SIMD and reads half the K / V fragments per MFMA)
  if constexpr (Elem<E>::bytes == 2) {
    if (option_or(OPT_ATTN_WAVES8, 1) != 0) {
      constexpr int WQ = DHP <= 64 ? 1 : 2;
      return a.dropout_p > 0.f ? launch_attn_bwd_v<E, DHP, true, 1, WQ>(a, s) : launch_attn_bwd_v<E, DHP, false, 1, WQ>(a, s);
    }
  }
  return a.dropout_p > 0.f ? launch_attn_bwd_v<E, DHP, true>(a, s) : launch_attn_bwd_v<E, DHP, false>(a, s);
}

}  // namespace dn

extern "C" int dn_attention_backward(const DnAttnBwdParams* pp, void* stream) {
  DN_CHECK_ARG(pp != nullptr, "dn_attention_backward: null params");
  const DnAttnBwdParams& p = *pp;
  DN_CHECK_ARG(p.q && p.k && p.v && p.out && p.dout && p.dq && p.dk && p.dv && p.lse && p.delta, "dn_attention_backward: null tensor");
  DN_CHECK_ARG(p.B > 0 && p.T > 0 && p.heads > 0 && p.dim_head > 0 && p.dim_head % 4 == 0, "dn_attention_backward: bad shape");
  DN_CHECK_ARG(p.dtype == DN_F32 || p.dtype == DN_BF16, "dn_attention_backward: bad dtype");
  DN_CHECK_ARG(p.dropout_p >= 0.f && p.dropout_p < 1.f, "dn_attention_backward: dropout_p %g", (double)p.dropout_p);
  const int es = p.dtype == DN_BF16 ? 2 : 4;
  DN_CHECK_ARG((p.dim_head * es) % 16 == 0, "dn_attention_backward: dim_head*elem must be a multiple of 16 bytes (dim_head=%d)", p.dim_head);
  for (int ld : {p.ldq, p.ldk, p.ldv, p.ldo, p.lddo, p.lddq, p.lddk, p.lddv})
    DN_CHECK_ARG((ld * es) % 16 == 0, "dn_attention_backward: row strides must keep 16-byte alignment");
  for (const void* t : {p.q, p.k, p.v, p.out, p.dout, (const void*)p.dq, (const void*)p.dk, (const void*)p.dv})
    DN_CHECK_ARG(((uintptr_t)t & 15) == 0, "dn_attention_backward: tensors must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  {
    const int64_t rows = (int64_t)p.B * p.T * p.heads;
    int G = 1;
    while (G < 64 && G * 8 < p.dim_head) G *= 2;  // lanes per (row, head)
    const int64_t per_block = 4 * (64 / G);
    hipLaunchKernelGGL(dn::attn_delta_kernel, dim3((unsigned)((rows + per_block - 1) / per_block)), dim3(256), 0, s, p.out, p.ldo, p.dout, p.lddo,
                       p.dtype, p.B, p.T, p.heads, p.dim_head, G, p.delta);
  }
  dn::AttnBwdArgs a;
  a.q = p.q; a.k = p.k; a.v = p.v; a.dout = p.dout; a.dq = p.dq; a.dk = p.dk; a.dv = p.dv;
  a.ldq = p.ldq; a.ldk = p.ldk; a.ldv = p.ldv; a.lddo = p.lddo; a.lddq = p.lddq; a.lddk = p.lddk; a.lddv = p.lddv;
  a.B = p.B; a.T = p.T; a.heads = p.heads; a.dim_head = p.dim_head; a.dtype = p.dtype;
  a.lengths = p.lengths; a.scale = p.scale; a.lse = p.lse; a.delta = p.delta;
  a.dropout_p = p.dropout_p; a.seed_lo = p.seed_lo; a.seed_hi = p.seed_hi;
  const int dh = p.dim_head;
  if (p.dtype == DN_BF16) {
    if (dh <= 32) return dn::launch_attn_bwd<dn::BF16, 32>(a, s);
    if (dh <= 64) return dn::launch_attn_bwd<dn::BF16, 64>(a, s);
    if (dh <= 96) return dn::launch_attn_bwd<dn::BF16, 96>(a, s);
    if (dh <= 128) return dn::launch_attn_bwd<dn::BF16, 128>(a, s);
  } else {
    if (dh <= 16) return dn::launch_attn_bwd<dn::F32, 16>(a, s);
    if (dh <= 32) return dn::launch_attn_bwd<dn::F32, 32>(a, s);
    if (dh <= 64) return dn::launch_attn_bwd<dn::F32, 64>(a, s);
    if (dh <= 96) return dn::launch_attn_bwd<dn::F32, 96>(a, s);
  }
  dn_set_error("dn_attention_backward: dim_head %d not supported", dh);
  return DN_EINVAL;
}
