// dn_attention: fused key-masked multi-head self-attention (flash-style, online softmax) for gfx950.
//
// Workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries (two 16-query
// tiles) and walks the keys in tiles of 64 staged in LDS.  Both products run "transposed" so that the
// query index lives on the MFMA column (= lane & 15):
//     S^T = K . Q^T      A operand = K rows from LDS (ds_read_b128), B operand = Q fragments in registers
//     O^T = V^T . P^T    A operand = V^T, B operand = P^T straight from the S^T accumulators
// With that orientation the softmax statistics of a query are lane-local up to two xor-shuffles
// (lanes l, l^16, l^32, l^48 hold the same query), the O rescale is a per-lane scalar, and P needs no
// LDS round trip: the S^T accumulator registers of two 16-key tiles, converted to bf16, ARE the B
// fragment of a 32-key k-step once the V^T fragment is gathered with the same key permutation --
// which is exactly what ds_read_b64_tr_b16 delivers from a row-major V tile (4 keys x 16 dims per
// 16-lane group).  In f32 mode each v_mfma_f32_16x16x4_f32 takes one P register and one scalar LDS
// read of V.  LDS rows are 256 B (bf16) / 512 B (f32) with an XOR swizzle that makes the K row reads,
// the V transposed reads and the f32 scalar reads bank-conflict-free (see lds_off).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace dn {

template <typename E> struct AttnGeom;
template <> struct AttnGeom<BF16> { static constexpr int ROWB = 256; };
template <> struct AttnGeom<F32> { static constexpr int ROWB = 512; };
template <> struct AttnGeom<F16> { static constexpr int ROWB = 256; };

// byte offset of byte `byte` (16-B aligned pieces stay contiguous) of row `row` in a K/V tile
template <typename E>
__device__ __forceinline__ int lds_off(int row, int byte);
template <>
__device__ __forceinline__ int lds_off<BF16>(int row, int byte) {
  return row * 256 + ((((byte >> 5) ^ (row & 7))) << 5) + (byte & 31);
}
template <>
__device__ __forceinline__ int lds_off<F16>(int row, int byte) { return lds_off<BF16>(row, byte); }
template <>
__device__ __forceinline__ int lds_off<F32>(int row, int byte) {
  return row * 512 + (((byte >> 4) ^ (row & 15)) << 4) + (byte & 15);
}

constexpr int KV_TILE = 64;
constexpr float NEG_BIG = -3.0e38f;

// QT: 16-query tiles per wave.  2 = four waves of 32 queries: a workgroup's waves sit on different SIMDs and two workgroups share a CU
// by their 64 KiB of LDS, i.e. two waves per SIMD at ~240 registers (one at 96 / 128-dim heads: 320-380 registers).  1 = eight waves
// of 16 queries: the same 128 queries and the same LDS per workgroup, half the accumulators and Q fragments per wave (106-122
// registers up to 64 dims, 150-161 above), so a CU holds FOUR waves per SIMD (two above 64 dims) -- more waves to cover the
// QK^T -> softmax -> PV dependency chain of each, at twice the K / V fragment reads per MFMA.  Bit-identical results (every query
// sees the same key tiles in the same order through the same MFMA shapes).  The 2-byte modes run QT = 1 (option attn_waves8 = 0:
// QT = 2; measured in the two-stream chain at [32,512], f16: 181.7 -> 183.5 steps/s, three alternating runs).
template <typename E, int DHP, bool DROP, int QT = 2>
__global__ __launch_bounds__(128 / (16 * QT) * 64, QT == 1 ? (DHP <= 64 ? 4 : 2) : 1) void attn_kernel(const DnAttnParams p) {
  constexpr int NT = 128 / (16 * QT) * 64;  // threads
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if constexpr (std::is_same<E, F16>::value) f16_saturate();
  constexpr int ES = Elem<E>::bytes;
  constexpr int ROWB = AttnGeom<E>::ROWB;
  constexpr int KS_D = DHP * ES / 64;    // 64-byte k-steps along the head dim (QK^T)
  constexpr int NCH = DHP * ES / 16;     // 16-byte chunks per K/V row
  constexpr int DT = DHP / 16;           // 16-wide output tiles along the head dim
  constexpr int TILE_LDS = KV_TILE * ROWB;
  char* k_lds = smem;                  // [2][KV_TILE][ROWB]
  char* v_lds = smem + 2 * TILE_LDS;   // [2][KV_TILE][ROWB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  int qblk, h, b;
  if (p.pad3_ & 1) { qblk = blockIdx.x; h = blockIdx.y; b = blockIdx.z; }  // A/B timing only (DN_ATTN_XCD=0): dispatch order
  else dn_xcd_block_map(qblk, h, b);
  const int T = p.T, dh = p.dim_head;
  const int Tk = p.Tk > 0 ? p.Tk : T;  // cross-attention: the keys are another sequence of Tk rows per batch element
  const int q0 = qblk * 128 + wave * (16 * QT);
  const int dhb = dh * ES;  // valid bytes per head row

  const char* qp = reinterpret_cast<const char*>(p.q) + ((int64_t)b * T * p.ldq + h * dh) * ES;
  const char* kp = reinterpret_cast<const char*>(p.k) + ((int64_t)b * Tk * p.ldk + h * dh) * ES;
  const char* vp = reinterpret_cast<const char*>(p.v) + ((int64_t)b * Tk * p.ldv + h * dh) * ES;

  // Q fragments (B operand): row = query, chunk = ks*4 + fg
  uint4 qf[QT][KS_D];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    int q = q0 + qt * 16 + fr;
    q = q < T ? q : T - 1;
#pragma unroll
    for (int ks = 0; ks < KS_D; ++ks) {
      const int byte = ks * 64 + fg * 16;
      qf[qt][ks] = byte < dhb ? *reinterpret_cast<const uint4*>(qp + (int64_t)q * p.ldq * ES + byte) : make_uint4(0, 0, 0, 0);
    }
  }

  f32x4 acc_o[DT][QT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) acc_o[i][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) { m_run[qt] = NEG_BIG; l_run[qt] = 0.f; }
  // bf16: the softmax denominators ride on the matrix pipe -- one extra output tile whose V^T fragment is all ones makes
  // acc_l[qt][r] = sum_k P[k][query] (the same bf16-rounded P the numerator uses), replacing 32 v_add + 2 cross-lane
  // reductions per key tile by 4 MFMAs on a pipe that has slack here (the loop is VALU-bound on the exponentials).
  f32x4 acc_l[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) acc_l[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr bool H16 = std::is_same<E, F16>::value;  // IEEE-half operands: same tiles and swizzles as bf16, P in [0, 1] rounds to 11 bits
  constexpr uint32_t ONE2 = H16 ? 0x3c003c00u : 0x3f803f80u;  // (1.0, 1.0) in the operand type
  const uint4 ones_frag = make_uint4(ONE2, ONE2, ONE2, ONE2);

  int len = p.lengths ? p.lengths[b] : Tk;
  len = len < Tk ? len : Tk;
  float sc = p.scale * 1.44269504088896340736f;  // work in log2 domain
  if (len <= 0) {  // every key masked: masked_fill makes all scores equal -> uniform softmax over all Tk keys
    len = Tk;
    sc = 0.f;
  }

  // attention dropout (training): thr = p * 2^32, kept probabilities scaled by 1 / (1 - p); the per-lane query rows' hashes are hoisted
  uint32_t drop_thr = 0, drop_row[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) drop_row[qt] = 0;
  float drop_inv = 1.f;
  if constexpr (DROP) {
    drop_thr = (uint32_t)fminf(p.dropout_p * 4294967296.0f, 4294967040.0f);
    drop_inv = 1.0f / (1.0f - p.dropout_p);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
      drop_row[qt] = dn_drop_row((uint32_t)(((int64_t)b * p.heads + h) * T + (q0 + qt * 16 + fr)), p.seed_lo);
  }
  auto dropped = [&](float pv, int qt, int key) -> float {
    if constexpr (DROP) return dn_drop_keep(drop_row[qt], (uint32_t)key, p.seed_hi, drop_thr) ? pv * drop_inv : 0.f;
    return pv;
  };

  // K/V tiles are double-buffered in LDS and register-staged TWO tiles ahead (two register sets): a key tile's work
  // (~1.5k cycles per wave) is shorter than a loaded L2/HBM round trip, so with one tile of lead every iteration ended
  // waiting for its successor's loads.  Tile t travels in register set t % 2: requested at the top of iteration t-2,
  // written to LDS buffer t % 2 at the end of iteration t-1 (that buffer was last read in iteration t-2; one barrier
  // per tile).  The loop is unrolled over two tiles so the set index is a compile-time constant.
  constexpr int NPT = (KV_TILE * NCH + NT - 1) / NT;  // 16-byte chunks per thread per tensor per tile
  uint4 kreg[2][NPT], vreg[2][NPT];
  auto issue_loads = [&](auto set_c, int kv0) {
    constexpr int S = decltype(set_c)::value;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int idx = tid + i * NT;
      const int row = idx / NCH, ch = idx - row * NCH;
      const int key = kv0 + row;
      kreg[S][i] = vreg[S][i] = make_uint4(0, 0, 0, 0);
      if (idx < KV_TILE * NCH && key < Tk && ch * 16 < dhb) {
        kreg[S][i] = *reinterpret_cast<const uint4*>(kp + (int64_t)key * p.ldk * ES + ch * 16);
        vreg[S][i] = *reinterpret_cast<const uint4*>(vp + (int64_t)key * p.ldv * ES + ch * 16);
      }
    }
  };
  auto write_tile = [&](auto set_c, int buf) {
    constexpr int S = decltype(set_c)::value;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int idx = tid + i * NT;
      const int row = idx / NCH, ch = idx - row * NCH;
      if (idx < KV_TILE * NCH) {
        const int off = buf * TILE_LDS + lds_off<E>(row, ch * 16);
        *reinterpret_cast<uint4*>(k_lds + off) = kreg[S][i];
        *reinterpret_cast<uint4*>(v_lds + off) = vreg[S][i];
      }
    }
  };
  // Per-lane LDS offsets of the fragment reads, hoisted out of the key loop: a tile row's swizzle key (row & 7, or & 15 for
  // f32) depends on the lane only (tile rows advance in multiples of 16), so every read below is one of these bases plus a
  // compile-time constant that folds into the ds_read offset field -- no address arithmetic per tile (the loop is
  // issue-bound on VALU/LDS instructions, not on the matrix pipe).
  int k_base[KS_D];  // row fr, 16-byte chunk ks*4 + fg
#pragma unroll
  for (int ks = 0; ks < KS_D; ++ks) k_base[ks] = lds_off<E>(fr, ks * 64 + fg * 16);
  int v_base[DT];    // bf16: row fg*4 + (fr >> 2), 8-byte piece dt*4 + (fr & 3)
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) v_base[dt] = lds_off<E>(fg * 4 + (fr >> 2), ES == 2 ? dt * 32 + (fr & 3) * 8 : 0);
  using SET0 = std::integral_constant<int, 0>;
  using SET1 = std::integral_constant<int, 1>;
  issue_loads(SET0{}, 0);
  if (KV_TILE < len) issue_loads(SET1{}, KV_TILE);
  write_tile(SET0{}, 0);
  __syncthreads();
  // one key tile: `cur_c` = its register-set parity (= its LDS buffer)
  auto key_tile = [&](auto cur_c, int kv0) {
    constexpr int buf = decltype(cur_c)::value;
    using NXT = std::integral_constant<int, 1 - buf>;
    const bool more = kv0 + KV_TILE < len;
    if (kv0 + 2 * KV_TILE < len) issue_loads(cur_c, kv0 + 2 * KV_TILE);  // this tile's set is free: it went to LDS last iteration
    const char* kt_lds = k_lds + buf * TILE_LDS;
    const char* vt_lds = v_lds + buf * TILE_LDS;

    // ---- S^T = K . Q^T : acc_s[kt][qt][r] = S[query qt*16+fr][key kt*16 + 4*fg + r]
    f32x4 acc_s[4][QT];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) acc_s[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS_D; ++ks) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const uint4 kf = *reinterpret_cast<const uint4*>(kt_lds + k_base[ks] + kt * 16 * ROWB);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) mma_kstep<E>(acc_s[kt][qt], kf, qf[qt][ks]);
      }
    }

    // ---- online softmax (log2 domain), per query tile.  The running max is kept on the raw scores (scaling by a
    // positive constant commutes with max), so the scale folds into one FMA per score: p = 2^(s*sc - m*sc); the
    // exponential is the bare v_exp_f32.  Key masking costs a compare + select per score and is compiled only into
    // the tile that straddles `len` (a wave-uniform branch).
    auto softmax_tile = [&](auto masked_tag) {
      constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float mx = NEG_BIG;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if constexpr (MASKED) {
              const int key = kv0 + kt * 16 + fg * 4 + r;
              acc_s[kt][qt][r] = key < len ? acc_s[kt][qt][r] : NEG_BIG;
            }
            mx = fmaxf(mx, acc_s[kt][qt][r]);
          }
        mx = quad_xor_max(mx);
        const float m_new = fmaxf(m_run[qt], mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_new) * sc);
        const float neg_ms = -m_new * sc;
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pv = __builtin_amdgcn_exp2f(fmaf(acc_s[kt][qt][r], sc, neg_ms));
            // a fully masked sequence runs with sc = 0 (uniform softmax over the Tk keys): the tile rows beyond Tk must still drop out
            if constexpr (MASKED) pv = (kv0 + kt * 16 + fg * 4 + r) < len ? pv : 0.f;
            acc_s[kt][qt][r] = pv;
            if constexpr (ES != 2) rs += pv;
          }
        if constexpr (ES != 2) {
          rs = quad_xor_sum(rs);
          l_run[qt] = l_run[qt] * alpha + rs;
        }
        // once the running maxima have settled (typically after the first key tiles) alpha is exactly 1 in every lane:
        // skip the rescale of the output tile then (wave-uniform test)
        if (__builtin_amdgcn_ballot_w64(m_new != m_run[qt]) != 0) {
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) acc_o[dt][qt] *= alpha;
          if constexpr (ES == 2) acc_l[qt] *= alpha;
        }
        m_run[qt] = m_new;
      }
    };
    if (kv0 + KV_TILE > len)
      softmax_tile(std::true_type{});
    else
      softmax_tile(std::false_type{});

    // ---- O^T += V^T . P^T
    if constexpr (ES == 2) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {  // 32-key k-steps = accumulator tiles (2kk, 2kk+1)
        uint4 pf[QT], pfd[QT];  // pf: the probabilities (denominator); pfd: after dropout (numerator)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          pf[qt].x = pack_h2<H16>(acc_s[2 * kk][qt][0], acc_s[2 * kk][qt][1]);
          pf[qt].y = pack_h2<H16>(acc_s[2 * kk][qt][2], acc_s[2 * kk][qt][3]);
          pf[qt].z = pack_h2<H16>(acc_s[2 * kk + 1][qt][0], acc_s[2 * kk + 1][qt][1]);
          pf[qt].w = pack_h2<H16>(acc_s[2 * kk + 1][qt][2], acc_s[2 * kk + 1][qt][3]);
          pfd[qt] = pf[qt];
          if constexpr (DROP) {
            const int ka = kv0 + (2 * kk) * 16 + fg * 4, kb = ka + 16;
            pfd[qt].x = pack_h2<H16>(dropped(acc_s[2 * kk][qt][0], qt, ka), dropped(acc_s[2 * kk][qt][1], qt, ka + 1));
            pfd[qt].y = pack_h2<H16>(dropped(acc_s[2 * kk][qt][2], qt, ka + 2), dropped(acc_s[2 * kk][qt][3], qt, ka + 3));
            pfd[qt].z = pack_h2<H16>(dropped(acc_s[2 * kk + 1][qt][0], qt, kb), dropped(acc_s[2 * kk + 1][qt][1], qt, kb + 1));
            pfd[qt].w = pack_h2<H16>(dropped(acc_s[2 * kk + 1][qt][2], qt, kb + 2), dropped(acc_s[2 * kk + 1][qt][3], qt, kb + 3));
          }
        }
        // transposed read: lane i of a 16-lane group supplies row (i>>2) cols 4*(i&3).., receives column i
        const int krow0 = (2 * kk) * 16 + fg * 4 + (fr >> 2);
        const int krow1 = krow0 + 16;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int byte = dt * 32 + (fr & 3) * 8;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(vt_lds + v_base[dt] + (2 * kk) * 16 * ROWB));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(vt_lds + v_base[dt] + (2 * kk + 1) * 16 * ROWB));
          uint4 vf;
          const uint2 lo2 = __builtin_bit_cast(uint2, lo), hi2 = __builtin_bit_cast(uint2, hi);
          vf.x = lo2.x; vf.y = lo2.y; vf.z = hi2.x; vf.w = hi2.y;
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) mma_kstep<E>(acc_o[dt][qt], vf, pfd[qt]);
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) mma_kstep<E>(acc_l[qt], ones_frag, pf[qt]);
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + fg * 4 + r;  // k-slot fg of MFMA (kt, r)
          float pq[QT];
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) pq[qt] = dropped(acc_s[kt][qt][r], qt, kv0 + key);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const float vv = *reinterpret_cast<const float*>(vt_lds + lds_off<E>(key, (dt * 16 + fr) * 4));
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) acc_o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, pq[qt], acc_o[dt][qt], 0, 0, 0);
          }
        }
    }
    if (more) write_tile(NXT{}, 1 - buf);  // the other buffer was last read one barrier ago
    __syncthreads();
  };
  for (int kv0 = 0; kv0 < len; kv0 += 2 * KV_TILE) {
    key_tile(SET0{}, kv0);
    if (kv0 + KV_TILE < len) key_tile(SET1{}, kv0 + KV_TILE);
  }

  // ---- O = acc / l ; lane holds dims dt*16 + 4*fg + 0..3 of query qt*16 + fr
  char* op = reinterpret_cast<char*>(p.out);
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = q0 + qt * 16 + fr;
    if (q >= T) continue;
    const float inv = 1.0f / (ES == 2 ? acc_l[qt][0] : l_run[qt]);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + fg * 4;
      if (d >= dh) continue;
      const f32x4 o = acc_o[dt][qt];
      store4(op, ((int64_t)b * T + q) * p.ldo + h * dh + d, p.dtype, o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
    }
    // log2-domain log-sum-exp of the scaled scores, kept for the backward pass: P = 2^(s * sc - lse)
    if (p.lse && fg == 0)
      p.lse[((int64_t)b * p.heads + h) * T + q] = m_run[qt] * sc + __builtin_amdgcn_logf(ES == 2 ? acc_l[qt][0] : l_run[qt]);
  }
}

// ------------------------------------------------------------------------------------------ split-operand (DN_BF16X3) attention
// q / k / v are plain fp32 in memory (the engine keeps them out of the split layout); every product runs as three bf16 MFMAs
// on (hi, lo) bf16 pairs made on the fly: Q is split once per workgroup into register fragments, a K / V tile is split when
// it goes from its staging registers to LDS -- a tile row holds the hi halves of the head's dims in bytes [0, 128) and the lo
// halves in [128, 256), i.e. the bf16 kernel's 256-byte rows and swizzle with "dims 64..127" = lo -- and P is split from the
// fp32 softmax output.  S^T = K_lo Q_hi + K_hi Q_lo + K_hi Q_hi, O^T += V_lo P_hi + V_hi P_lo + V_hi P_hi; softmax, masks
// and the denominators are fp32 as in the exact-fp32 kernel.  96 bf16 MFMAs per 64-key tile instead of 256 fp32 ones.
// dim_head <= 64 (larger heads run the exact-fp32 kernel).
// QT: 16-query tiles per wave (2 = four waves of 32 queries, 1 = eight waves of 16: attn_kernel's note)
template <int DHP, bool DROP, int QT = 2>
__global__ __launch_bounds__(128 / (16 * QT) * 64, QT == 1 ? 4 : 2) void attn_x3_kernel(const DnAttnParams p) {
  constexpr int NT = 128 / (16 * QT) * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(DHP == 32 || DHP == 64, "padded head dim");
  using L = BF16;                     // LDS geometry and swizzle of the bf16 kernel
  constexpr int ROWB = AttnGeom<L>::ROWB, LO = 128;
  constexpr int KS_D = DHP / 32;      // 32-dim k-steps of QK^T
  constexpr int NCH = DHP / 4;        // 16-byte (4 x fp32) chunks per K / V row in memory
  constexpr int DT = DHP / 16;
  constexpr int TILE_LDS = KV_TILE * ROWB;
  char* k_lds = smem;
  char* v_lds = smem + 2 * TILE_LDS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  int qblk, h, b;
  if (p.pad3_ & 1) { qblk = blockIdx.x; h = blockIdx.y; b = blockIdx.z; }
  else dn_xcd_block_map(qblk, h, b);
  const int T = p.T, dh = p.dim_head;
  const int Tk = p.Tk > 0 ? p.Tk : T;
  const int q0 = qblk * 128 + wave * (16 * QT);
  const float* qp = reinterpret_cast<const float*>(p.q) + (int64_t)b * T * p.ldq + h * dh;
  const float* kp = reinterpret_cast<const float*>(p.k) + (int64_t)b * Tk * p.ldk + h * dh;
  const float* vp = reinterpret_cast<const float*>(p.v) + (int64_t)b * Tk * p.ldv + h * dh;

  auto split4 = [](const float4 v, uint2& hi, uint2& lo) {
    split_pair(v.x, v.y, hi.x, lo.x);
    split_pair(v.z, v.w, hi.y, lo.y);
  };
  // Q fragments (B operand): lane (fr, fg) holds dims ks*32 + fg*8 .. +7 of query fr, as hi and lo halves
  uint4 qh[QT][KS_D], ql[QT][KS_D];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    int q = q0 + qt * 16 + fr;
    q = q < T ? q : T - 1;
#pragma unroll
    for (int ks = 0; ks < KS_D; ++ks) {
      const int d = ks * 32 + fg * 8;
      const float* src = qp + (int64_t)q * p.ldq + d;
      const float4 a = d < dh ? *reinterpret_cast<const float4*>(src) : make_float4(0, 0, 0, 0);
      const float4 c = d + 4 < dh ? *reinterpret_cast<const float4*>(src + 4) : make_float4(0, 0, 0, 0);
      uint2 h0, l0, h1, l1;
      split4(a, h0, l0);
      split4(c, h1, l1);
      qh[qt][ks] = make_uint4(h0.x, h0.y, h1.x, h1.y);
      ql[qt][ks] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
  }

  f32x4 acc_o[DT][QT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) acc_o[i][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) { m_run[qt] = NEG_BIG; l_run[qt] = 0.f; }

  int len = p.lengths ? p.lengths[b] : Tk;
  len = len < Tk ? len : Tk;
  float sc = p.scale * 1.44269504088896340736f;
  if (len <= 0) {
    len = Tk;
    sc = 0.f;
  }
  uint32_t drop_thr = 0, drop_row[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) drop_row[qt] = 0;
  float drop_inv = 1.f;
  if constexpr (DROP) {
    drop_thr = (uint32_t)fminf(p.dropout_p * 4294967296.0f, 4294967040.0f);
    drop_inv = 1.0f / (1.0f - p.dropout_p);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
      drop_row[qt] = dn_drop_row((uint32_t)(((int64_t)b * p.heads + h) * T + (q0 + qt * 16 + fr)), p.seed_lo);
  }
  auto dropped = [&](float pv, int qt, int key) -> float {
    if constexpr (DROP) return dn_drop_keep(drop_row[qt], (uint32_t)key, p.seed_hi, drop_thr) ? pv * drop_inv : 0.f;
    return pv;
  };

  // fp32 K / V tiles travel global -> registers ONE tile ahead (a tile's three-product work, ~2.5k cycles, covers a loaded L2
  // round trip; a second register set would cost the second workgroup per CU) and are split into bf16 halves on the way to LDS
  constexpr int NPT = KV_TILE * NCH / NT;  // 16-byte chunks per thread per tensor per tile
  constexpr int RSTEP = NT / NCH;          // tile rows between a thread's chunks: its chunk column is the same in all of them
  static_assert(RSTEP % 8 == 0, "a thread's rows share one swizzle key");
  const int ld_r0 = tid / NCH, ld_ch = tid - ld_r0 * NCH;
  const bool ld_col = ld_ch * 4 < dh;
  const float* kp_t = kp + (int64_t)ld_r0 * p.ldk + ld_ch * 4;  // per-thread bases; the tile / chunk row offsets are uniform
  const float* vp_t = vp + (int64_t)ld_r0 * p.ldv + ld_ch * 4;
  const int wr_h = lds_off<L>(ld_r0, ld_ch * 8), wr_l = lds_off<L>(ld_r0, LO + ld_ch * 8);
  // ONE staging register set serves both tensors of the next tile: its K rows are requested at the top of a tile and written
  // to the other LDS buffer (free for the whole tile) behind the S^T products, its V rows are requested then and written at the
  // end -- half the staging registers, which is what lets two workgroups share a CU without spills.
  float4 sreg[NPT];
  auto issue_loads = [&](const float* base, int ld, int kv0) {
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int key = kv0 + i * RSTEP;  // + ld_r0
      sreg[i] = make_float4(0, 0, 0, 0);
      if (ld_col && key + ld_r0 < Tk) sreg[i] = *reinterpret_cast<const float4*>(base + (int64_t)key * ld);
    }
  };
  auto write_tile = [&](char* lds, int buf) {
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int o = buf * TILE_LDS + i * RSTEP * ROWB;
      uint2 hi, lo;
      split4(sreg[i], hi, lo);
      *reinterpret_cast<uint2*>(lds + wr_h + o) = hi;
      *reinterpret_cast<uint2*>(lds + wr_l + o) = lo;
    }
  };
  int k_base_h[KS_D], k_base_l[KS_D], v_base_h[DT], v_base_l[DT];
#pragma unroll
  for (int ks = 0; ks < KS_D; ++ks) {
    k_base_h[ks] = lds_off<L>(fr, ks * 64 + fg * 16);
    k_base_l[ks] = lds_off<L>(fr, LO + ks * 64 + fg * 16);
  }
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    v_base_h[dt] = lds_off<L>(fg * 4 + (fr >> 2), dt * 32 + (fr & 3) * 8);
    v_base_l[dt] = lds_off<L>(fg * 4 + (fr >> 2), LO + dt * 32 + (fr & 3) * 8);
  }
  using SET0 = std::integral_constant<int, 0>;
  using SET1 = std::integral_constant<int, 1>;
  issue_loads(kp_t, p.ldk, 0);
  write_tile(k_lds, 0);
  issue_loads(vp_t, p.ldv, 0);
  write_tile(v_lds, 0);
  __syncthreads();
  auto tr_read = [&](const char* base) -> uint2 {
    const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
    return __builtin_bit_cast(uint2, r);
  };
  auto key_tile = [&](auto cur_c, int kv0) {
    constexpr int buf = decltype(cur_c)::value;
    const bool more = kv0 + KV_TILE < len;
    if (more) issue_loads(kp_t, p.ldk, kv0 + KV_TILE);  // the next tile's K rows land under this tile's S^T products
    const char* kt_lds = k_lds + buf * TILE_LDS;
    const char* vt_lds = v_lds + buf * TILE_LDS;

    // ---- S^T = K . Q^T, three bf16 products, small terms first
    f32x4 acc_s[4][QT];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) acc_s[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS_D; ++ks) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const uint4 kh = *reinterpret_cast<const uint4*>(kt_lds + k_base_h[ks] + kt * 16 * ROWB);
        const uint4 kl = *reinterpret_cast<const uint4*>(kt_lds + k_base_l[ks] + kt * 16 * ROWB);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) mma_kstep<BF16>(acc_s[kt][qt], kl, qh[qt][ks]);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) mma_kstep<BF16>(acc_s[kt][qt], kh, ql[qt][ks]);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) mma_kstep<BF16>(acc_s[kt][qt], kh, qh[qt][ks]);
      }
    }
    if (more) {  // K rows to the other buffer (last read one barrier ago); the same registers then fetch the V rows
      write_tile(k_lds, 1 - buf);
      issue_loads(vp_t, p.ldv, kv0 + KV_TILE);
    }
    // ---- online softmax in fp32 (log2 domain), as the other kernels
    auto softmax_tile = [&](auto masked_tag) {
      constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float mx = NEG_BIG;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if constexpr (MASKED) {
              const int key = kv0 + kt * 16 + fg * 4 + r;
              acc_s[kt][qt][r] = key < len ? acc_s[kt][qt][r] : NEG_BIG;
            }
            mx = fmaxf(mx, acc_s[kt][qt][r]);
          }
        mx = quad_xor_max(mx);
        const float m_new = fmaxf(m_run[qt], mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_new) * sc);
        const float neg_ms = -m_new * sc;
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pv = __builtin_amdgcn_exp2f(fmaf(acc_s[kt][qt][r], sc, neg_ms));
            // a fully masked sequence runs with sc = 0 (uniform softmax over the Tk keys): the tile rows beyond Tk must still drop out
            if constexpr (MASKED) pv = (kv0 + kt * 16 + fg * 4 + r) < len ? pv : 0.f;
            acc_s[kt][qt][r] = pv;
            rs += pv;
          }
        rs = quad_xor_sum(rs);
        l_run[qt] = l_run[qt] * alpha + rs;
        if (__builtin_amdgcn_ballot_w64(m_new != m_run[qt]) != 0) {
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) acc_o[dt][qt] *= alpha;
        }
        m_run[qt] = m_new;
      }
    };
    if (kv0 + KV_TILE > len) softmax_tile(std::true_type{});
    else softmax_tile(std::false_type{});

    // ---- O^T += V^T . P^T with P split into its bf16 halves
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 ph[QT], pl[QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const int ka = kv0 + (2 * kk) * 16 + fg * 4, kb = ka + 16;
        float v8[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v8[r] = dropped(acc_s[2 * kk][qt][r], qt, ka + r);
          v8[4 + r] = dropped(acc_s[2 * kk + 1][qt][r], qt, kb + r);
        }
        split_pair(v8[0], v8[1], ph[qt].x, pl[qt].x);
        split_pair(v8[2], v8[3], ph[qt].y, pl[qt].y);
        split_pair(v8[4], v8[5], ph[qt].z, pl[qt].z);
        split_pair(v8[6], v8[7], ph[qt].w, pl[qt].w);
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const uint2 h0 = tr_read(vt_lds + v_base_h[dt] + (2 * kk) * 16 * ROWB), h1 = tr_read(vt_lds + v_base_h[dt] + (2 * kk + 1) * 16 * ROWB);
        const uint2 l0 = tr_read(vt_lds + v_base_l[dt] + (2 * kk) * 16 * ROWB), l1 = tr_read(vt_lds + v_base_l[dt] + (2 * kk + 1) * 16 * ROWB);
        const uint4 vh = make_uint4(h0.x, h0.y, h1.x, h1.y), vl = make_uint4(l0.x, l0.y, l1.x, l1.y);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) mma_kstep<BF16>(acc_o[dt][qt], vl, ph[qt]);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) mma_kstep<BF16>(acc_o[dt][qt], vh, pl[qt]);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) mma_kstep<BF16>(acc_o[dt][qt], vh, ph[qt]);
      }
    }
    if (more) write_tile(v_lds, 1 - buf);
    __syncthreads();
  };
  for (int kv0 = 0; kv0 < len; kv0 += 2 * KV_TILE) {
    key_tile(SET0{}, kv0);
    if (kv0 + KV_TILE < len) key_tile(SET1{}, kv0 + KV_TILE);
  }

  char* op = reinterpret_cast<char*>(p.out);
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = q0 + qt * 16 + fr;
    if (q >= T) continue;
    const float inv = 1.0f / l_run[qt];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + fg * 4;
      if (d >= dh) continue;
      const f32x4 o = acc_o[dt][qt];
      store4(op, ((int64_t)b * T + q) * p.ldo + h * dh + d, p.dtype, o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
    }
    if (p.lse && fg == 0) p.lse[((int64_t)b * p.heads + h) * T + q] = m_run[qt] * sc + __builtin_amdgcn_logf(l_run[qt]);
  }
}

template <int DHP, bool DROP, int QT = 2>
static int launch_attn_x3_v(const DnAttnParams& p, hipStream_t s) {
  constexpr int lds = 4 * KV_TILE * AttnGeom<BF16>::ROWB;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_x3_kernel<DHP, DROP, QT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  dim3 grid((p.T + 127) / 128, p.heads, p.B);
  hipLaunchKernelGGL((attn_x3_kernel<DHP, DROP, QT>), grid, dim3(128 / (16 * QT) * 64), lds, s, p);
  DN_CHECK_LAUNCH("dn_attention (split operands)");
  return DN_OK;
}
template <int DHP>
static int launch_attn_x3(const DnAttnParams& p, hipStream_t s) {
  if (option_or(OPT_ATTN_WAVES8, 1) != 0)
    return p.dropout_p > 0.f ? launch_attn_x3_v<DHP, true, 1>(p, s) : launch_attn_x3_v<DHP, false, 1>(p, s);
  return p.dropout_p > 0.f ? launch_attn_x3_v<DHP, true>(p, s) : launch_attn_x3_v<DHP, false>(p, s);
}

template <typename E, int DHP, bool DROP, int QT = 2>
static int launch_attn_v(const DnAttnParams& p, hipStream_t s) {
  constexpr int lds = 4 * KV_TILE * AttnGeom<E>::ROWB;  // K and V, double-buffered
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kernel<E, DHP, DROP, QT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  dim3 grid((p.T + 127) / 128, p.heads, p.B);
  hipLaunchKernelGGL((attn_kernel<E, DHP, DROP, QT>), grid, dim3(128 / (16 * QT) * 64), lds, s, p);
  DN_CHECK_LAUNCH("dn_attention");
  return DN_OK;
}

template <typename E, int DHP>
static int launch_attn(const DnAttnParams& p, hipStream_t s) {
  if constexpr (Elem<E>::bytes == 2) {
    if (option_or(OPT_ATTN_WAVES8, 1) != 0)
      return p.dropout_p > 0.f ? launch_attn_v<E, DHP, true, 1>(p, s) : launch_attn_v<E, DHP, false, 1>(p, s);
  }
  return p.dropout_p > 0.f ? launch_attn_v<E, DHP, true>(p, s) : launch_attn_v<E, DHP, false>(p, s);
}
template <int DHP>
static int launch_attn_f16(const DnAttnParams& p, hipStream_t s) {  // inference mode: no dropout
  return option_or(OPT_ATTN_WAVES8, 1) != 0 ? launch_attn_v<F16, DHP, false, 1>(p, s) : launch_attn_v<F16, DHP, false>(p, s);
}

}  // namespace dn

extern "C" int dn_attention(const DnAttnParams* pp, void* stream) {
  DN_CHECK_ARG(pp != nullptr, "dn_attention: null params");
  DnAttnParams p = *pp;
  static const int xcd_off = getenv("DN_ATTN_XCD") && atoi(getenv("DN_ATTN_XCD")) == 0 ? 1 : 0;  // A/B timing knob, read once
  p.pad3_ = xcd_off;
  DN_CHECK_ARG(p.q && p.k && p.v && p.out, "dn_attention: null tensor");
  DN_CHECK_ARG(p.B > 0 && p.T > 0 && p.heads > 0 && p.dim_head > 0 && p.Tk >= 0, "dn_attention: bad shape");
  DN_CHECK_ARG(!(p.lse && p.Tk > 0 && p.Tk != p.T), "dn_attention: the backward pass (lse) covers self-attention only");
  // DN_BF16X3: q / k / v are plain fp32 (the engine keeps them out of the split layout); the products run as three bf16 MFMAs on
  // halves split on the fly (attn_x3_kernel; heads over 64 dims: exact fp32 MFMA) and only the output -- the operand of the to_out
  // contraction -- is written as split rows (store4 on p.dtype)
  DN_CHECK_ARG(p.dtype == DN_F32 || p.dtype == DN_BF16 || p.dtype == DN_BF16X3 || p.dtype == DN_F16, "dn_attention: bad dtype");
  if (p.dtype == DN_BF16X3) DN_CHECK_ARG(p.ldo % 32 == 0 && (p.heads * p.dim_head) % 4 == 0 && ((uintptr_t)p.out & 127) == 0, "dn_attention: split-row output needs ldo %% 32 == 0");
  DN_CHECK_ARG(p.dropout_p >= 0.f && p.dropout_p < 1.f, "dn_attention: dropout_p %g", (double)p.dropout_p);
  const int es = dn::dn_is16(p.dtype) ? 2 : 4;
  DN_CHECK_ARG((p.dim_head * es) % 16 == 0, "dn_attention: dim_head*elem must be a multiple of 16 bytes (dim_head=%d)", p.dim_head);
  DN_CHECK_ARG((p.ldq * es) % 16 == 0 && (p.ldk * es) % 16 == 0 && (p.ldv * es) % 16 == 0 && p.ldo % 4 == 0,
               "dn_attention: row strides must keep 16-byte alignment");
  DN_CHECK_ARG(((uintptr_t)p.q & 15) == 0 && ((uintptr_t)p.k & 15) == 0 && ((uintptr_t)p.v & 15) == 0 && ((uintptr_t)p.out & 15) == 0,
               "dn_attention: tensors must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int dh = p.dim_head;
  if (p.dtype == DN_BF16X3) {  // heads up to 64 dims: three bf16 products on the fly; larger heads: the exact-fp32 kernel
    static const bool x3_off = getenv("DN_ATTN_X3") && atoi(getenv("DN_ATTN_X3")) == 0;  // A/B: exact-fp32 products for every head size
    if (!x3_off && dh <= 32) return dn::launch_attn_x3<32>(p, s);
    if (!x3_off && dh <= 64) return dn::launch_attn_x3<64>(p, s);
  }
  if (p.dtype == DN_BF16) {
    if (dh <= 32) return dn::launch_attn<dn::BF16, 32>(p, s);
    if (dh <= 64) return dn::launch_attn<dn::BF16, 64>(p, s);
    if (dh <= 96) return dn::launch_attn<dn::BF16, 96>(p, s);
    if (dh <= 128) return dn::launch_attn<dn::BF16, 128>(p, s);
  } else if (p.dtype == DN_F16) {
    DN_CHECK_ARG(p.dropout_p == 0.f && !p.lse, "dn_attention: DN_F16 is an inference mode (no dropout, no saved log-sum-exp)");
    if (dh <= 32) return dn::launch_attn_f16<32>(p, s);
    if (dh <= 64) return dn::launch_attn_f16<64>(p, s);
    if (dh <= 96) return dn::launch_attn_f16<96>(p, s);
    if (dh <= 128) return dn::launch_attn_f16<128>(p, s);
  } else {
    if (dh <= 16) return dn::launch_attn<dn::F32, 16>(p, s);
    if (dh <= 32) return dn::launch_attn<dn::F32, 32>(p, s);
    if (dh <= 64) return dn::launch_attn<dn::F32, 64>(p, s);
    if (dh <= 96) return dn::launch_attn<dn::F32, 96>(p, s);
  }
  dn_set_error("dn_attention: dim_head %d not supported", dh);
  return DN_EINVAL;
}
