// dn_conv_gemm: causal-conv / linear contraction on MFMA with fused epilogues (gfx950).
//
//   out[g][m, n] = epi( sum_{term} sum_k A_term[g][m - shift_term, k] * W_term[g][n, k] )
//
// A workgroup owns a BM(activation rows) x 128(weight rows) output tile, BM = 256 (8 waves) or 128
// (4 waves), each wave a 64 x 64 sub-tile; the K loop walks (term, 128-byte K-tile) pairs through an LDS
// ring (3 stages of 48 KiB at BM = 256).  Both operand tiles are staged L2/HBM -> LDS by LDS-DMA
// (global_load_lds_dwordx4, 1 KiB per wave-instruction); the LDS image is linear in lane order and the
// bank-conflict swizzle (16-byte chunk ^= row & 7) is applied on the per-lane SOURCE address and again
// on the ds_read_b128 address.  Frames before the start of a sequence (t < shift) are sourced from a
// 128-byte zero page, so activations stay dense [M, ld] with no per-sequence padding.  Weights are the
// MFMA A operand and activations the B operand, so each lane ends up with 4 consecutive output columns
// of one row: bias/FiLM vectors load as float4 and stores are 8/16 bytes per lane.
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#pragma once
#include "common.h"

// Ablation switches of the two-waves-per-SIMD K loops (tools/gemm_bench.py): compile-time, because a run-time test inside
// the loop costs the wave that feeds the MFMA pipe a branch per test.  Build with EXTRA=-DDN_GEMM_ABL=<bits>:
// bit0 skip DMA, bit1 skip MFMA, bit2 skip LDS fragment reads, bit3 no s_setprio.
#ifndef DN_GEMM_ABL
#define DN_GEMM_ABL 0
#endif

namespace dn {

static __device__ uint4 g_zero_page[8];  // 128 bytes of zeros (static storage is zero-initialised)

constexpr int BN = 128, ROWB = 128;   // weight rows per tile; bytes of K per row per K-tile
constexpr int W_TILE_BYTES = BN * ROWB;  // 16 KiB

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA of 16 B per lane: lane l's bytes land at LDS address `lds_addr` + 16*l (wave-uniform base in M0).
// Issued through inline asm on purpose: hipcc cannot tell which LDS bytes a global_load_lds writes, so with
// the builtin it drains vmcnt(0) before the first ds_read of the tile being consumed and the prefetch never
// overlaps the MFMAs.  The asm form is invisible to its waitcnt pass; completion is ordered by the counted
// s_waitcnt vmcnt(N) + s_barrier at the top of the K loop (pipe_sync).
__device__ __forceinline__ void glds16(const void* src, uint32_t lds_addr) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_addr))  // (uniform; the readfirstlane folds away unless the compiler chose a VGPR for it)
      : "memory");
}

// Same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset (saddr form): pieces that differ only
// by a uniform row offset share one offset VGPR.
__device__ __forceinline__ void glds16_s(uint32_t voff, uint64_t sbase, uint32_t lds_addr) {
  uint32_t keep;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sbase), hi = __builtin_amdgcn_readfirstlane((uint32_t)(sbase >> 32));
  const uint64_t base = ((uint64_t)hi << 32) | lo;
  asm volatile(
      // s_nop 3: with the two s_movs, 5+ wait states between a v_readfirstlane that produced `base` and the VMEM read of it
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(base), "s"(lds_addr)
      : "memory");
}

// The same DMA in two statements, for streams that place them in different MFMA gaps: M0 <- LDS address (one wait state
// is needed before the DMA reads it; the instruction the caller puts in between provides it), then the load.  M0 is
// declared clobbered instead of being saved and restored (nothing else in such a loop uses it).
__device__ __forceinline__ void glds_set_m0(uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
}
__device__ __forceinline__ void glds_go(const void* src) {
  asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(src) : "memory");
}
// The same load from lanes 0..15 only (a 256-byte sub-piece: 4 rows of a 64-byte K-tile); EXEC is restored inside the statement.
__device__ __forceinline__ void glds_go_lanes16(const void* src) {
  uint64_t keep;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 0xffff\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b64 exec, %0"
               : "=&s"(keep)
               : "v"(src)
               : "memory");
}
// glds16 from the lanes of `mask` only (a sub-piece of fewer than 16 rows).
__device__ __forceinline__ void glds16_lanes(const void* src, uint32_t lds_addr, uint64_t mask) {
  uint32_t keep;
  uint64_t keepx;
  const uint32_t ml = __builtin_amdgcn_readfirstlane((uint32_t)mask), mh = __builtin_amdgcn_readfirstlane((uint32_t)(mask >> 32));
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, %4\n\tglobal_load_lds_dwordx4 %2, off\n\t"
      "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
      : "=&s"(keep), "=&s"(keepx)
      : "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_addr)), "s"(((uint64_t)mh << 32) | ml)
      : "memory");
}
// saddr form.  Hazard the compiler cannot see through inline asm: an SGPR written by a VALU instruction
// (v_readfirstlane, which is how a uniform value computed in VGPRs reaches an "s" operand) needs 5 wait states before
// a VMEM instruction reads it.  So the base is first copied by a SALU instruction -- in the M0 statement, one MFMA gap
// ahead of the load -- and the load reads the copy (SALU-written SGPRs are interlocked).
__device__ __forceinline__ uint64_t glds_set_m0_base(uint32_t lds_addr, uint64_t sbase) {
  uint64_t copy;
  // (wave-uniform values the compiler happens to hold in VGPRs reach the SALU moves through v_readfirstlane)
  const uint32_t la = __builtin_amdgcn_readfirstlane(lds_addr);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sbase), hi = __builtin_amdgcn_readfirstlane((uint32_t)(sbase >> 32));
  const uint64_t base = ((uint64_t)hi << 32) | lo;
  asm volatile("s_mov_b32 m0, %1\n\ts_mov_b64 %0, %2" : "=&s"(copy) : "s"(la), "s"(base) : "memory");
  return copy;
}
__device__ __forceinline__ void glds_go_s(uint32_t voff, uint64_t sbase_copy) {
  asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase_copy) : "memory");
}

// wait until at most N of this wave's LDS-DMA pieces are outstanding, then workgroup barrier; one asm
// statement with a memory clobber so no ds_read of the tile is scheduled above it.
template <int N>
__device__ __forceinline__ void pipe_sync() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// Row-wise epilogue of one wave's 64 x 64 fp32 slab `ep` ([64][EP_LD] floats in LDS): slab row r is output row
// m_base + r, slab column c is packed weight row n_base + c (n_base a multiple of 64).  16 lanes cover the 64
// columns of a row, so bias / FiLM / residual loads and the output stores are whole 128-256 B row segments.
constexpr int EP_LD = 68;

// Logical tile index -> (row tile, column tile).  band = 1: column tiles fastest (the tiles that share one activation row panel
// are neighbours: right when the weights fit the L2 and the activations stream).  band > 1 (launch(): weights much larger than
// an L2): bands of `band` row tiles, inside a band row tiles fastest -- the tiles an XCD runs together then form a
// (band x concurrency / band) block, so each weight column panel serves `band` row tiles per fetch instead of one.
__device__ __forceinline__ void tile_coords(int logical, int n_tiles_n, int m_tiles, int band, int& mt, int& nt) {
  if (band <= 1) {
    mt = logical / n_tiles_n;
    nt = logical - mt * n_tiles_n;
    return;
  }
  const int per_band = band * n_tiles_n;
  const int b = logical / per_band, r = logical - b * per_band;
  const int rows = min(band, m_tiles - b * band);  // the last band may be short
  nt = r / rows;
  mt = b * band + (r - nt * rows);
}

// KIND: 0 = fp32, 1 = bf16, 2 = DN_BF16X3 split rows (hi / lo bf16 halves of a 32-element group, common.h), 3 = IEEE half
template <int KIND>
__device__ __forceinline__ void store4t(void* base, int64_t off, float a, float b, float c, float d) {
  if constexpr (KIND == 2)
    store4_split(base, off, a, b, c, d);
  else if constexpr (KIND == 1 || KIND == 3)
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + off) = make_uint2(pack_h2<KIND == 3>(a, b), pack_h2<KIND == 3>(c, d));
  else
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + off) = make_float4(a, b, c, d);
}
template <int KIND>  // 0 = fp32, 1 = bf16, 3 = IEEE half
__device__ __forceinline__ float4 load4t(const void* base, int64_t off) {
  if constexpr (KIND != 0) {
    const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + off);
    const float2 a = unpack_h2<KIND == 3>(v.x), b = unpack_h2<KIND == 3>(v.y);
    return make_float4(a.x, a.y, b.x, b.y);
  } else {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
  }
}

// OUT_BF / RES_BF: storage types fixed at compile time; FULL: every row of the slab is inside M (no per-row guards).
// Split RMSNorm, consumer side (DnGemmParams.row_ssq): lane r of the wave turns the partial sums of squares of row
// m_base + r into the row's factor sqrt(D) / max(|row|, 1e-12) and parks it in the slab's pad column 64, where the
// epilogue's row loop picks it up (one LDS pipeline per wave, in order: no barrier needed).
__device__ __forceinline__ void park_row_scales(const DnGemmParams& p, float* ep, int m_base, int lane) {
  int m = m_base + lane;
  m = m < p.M ? m : p.M - 1;
  const float* q = p.row_ssq + (int64_t)m * p.row_ssq_ld;
  float ss = 0.f;
  for (int j = 0; j < p.row_ssq_parts; ++j) ss += q[j];  // same order as row_scale_finish
  ep[lane * EP_LD + 64] = sqrtf(p.row_D) / fmaxf(sqrtf(ss), 1e-12f);
}
// The same factor with its loads taken out of the epilogue: a kernel requests the (up to 8) partials of "its" row
// (lane r <-> row m_base + r of the wave's slab) before it starts staging, and finishes the arithmetic after the staging
// prologue's first counted vmcnt -- VMEM loads return in order and these are older than every DMA piece, so they have
// landed by then without a wait of their own.  Inline asm, because the compiler cannot see the DMA pieces and would
// drain vmcnt(0) (the whole staging prologue) in front of the first use of an ordinary load.
struct RowSsqReq { f32x4 a, b; bool on; };
template <int EPI>
__device__ __forceinline__ RowSsqReq row_scale_request(const DnGemmParams& p, int m_base, int lane) {
  RowSsqReq r;
  r.a = r.b = f32x4{0.f, 0.f, 0.f, 0.f};
  r.on = false;
  if constexpr (EPI == DN_EPI_BIAS || EPI == DN_EPI_SILU || EPI == DN_EPI_GEGLU) {
    r.on = p.row_ssq != nullptr && p.row_ssq_parts <= 8 && (p.row_ssq_ld & 3) == 0 && p.row_ssq_ld >= 8;
    if (r.on) {
      int m = m_base + lane;
      m = m < p.M ? m : p.M - 1;
      const float* q = p.row_ssq + (int64_t)m * p.row_ssq_ld;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r.a) : "v"(q) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r.b) : "v"(q + 4) : "memory");
    }
  }
  return r;
}
// call after a counted vmcnt that covers the request; < 0: not requested (the epilogue loads the partials itself)
__device__ __forceinline__ float row_scale_finish(const DnGemmParams& p, RowSsqReq r) {
  if (!r.on) return -1.f;
  asm volatile("" : "+v"(r.a), "+v"(r.b));  // ordered behind the preceding (volatile) vmcnt statement
  const float v[8] = {r.a[0], r.a[1], r.a[2], r.a[3], r.b[0], r.b[1], r.b[2], r.b[3]};
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) ss += j < p.row_ssq_parts ? v[j] : 0.f;
  return sqrtf(p.row_D) / fmaxf(sqrtf(ss), 1e-12f);
}
// sum over the 16 lanes of a DPP row (lanes 16 j .. 16 j + 15), result in every lane: xor-1 and xor-2 butterflies inside a
// quad, then the mirrored half-row and the mirrored row -- four VALU adds, no LDS crossbar latency
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float q) {
  q += dpp_mov<0xB1>(q);   // quad_perm [1,0,3,2]
  q += dpp_mov<0x4E>(q);   // quad_perm [2,3,0,1]
  q += dpp_mov<0x141>(q);  // row_half_mirror
  q += dpp_mov<0x140>(q);  // row_mirror
  return q;
}
// The four row factors a lane needs when it copies its accumulators to the transpose slab (rows 16 mt + (lane & 15)),
// from the per-lane factors (lane r <-> row r) of row_scale_finish.  < 0 in, 1.0 out (nothing pre-scaled).
constexpr float ROW_PRESCALED = -2.f;
__device__ __forceinline__ f32x4 scaled(const f32x4& a, float sc) { return f32x4{a[0] * sc, a[1] * sc, a[2] * sc, a[3] * sc}; }
__device__ __forceinline__ void slab_row_scales(float row_scale, int lane, float (&sc)[4]) {
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) sc[mt] = row_scale >= 0.f ? __shfl(row_scale, mt * 16 + (lane & 15)) : 1.f;
}
__device__ __forceinline__ const float* row_bias_of(const DnGemmParams& p, int m) {
  return p.row_bias + (p.row_bias_ld ? (int64_t)(m / p.T) * p.row_bias_ld : 0);
}

// 8 consecutive bf16 outputs per lane = one 16-byte store: bf16 output stores are issue-bound (a wave-instruction moves
// 512 B as dwordx2 but 1 KiB as dwordx4), so halving their number halves the store tail of the epilogue.
// Is the frame a term reads for output frame t inside the sequence?  shift >= 0: the causal case, frame t - shift (frames before the
// sequence start are zeros); shift < 0: the transposed conv of the backward data path, frame t + |shift| (zeros past the end).
__device__ __forceinline__ bool shift_valid(int t, int shift, int T) { return shift >= 0 ? t >= shift : t - shift < T; }

// Element offset of output (row m, column n): row-major, or K-blocked [N/32][M][32] for a consumer that stages whole
// cache lines (DN_LAYOUT_OUT_KBLOCKED; the 4 or 8 columns a lane stores never straddle a 32-column block).
__device__ __forceinline__ int64_t out_off(const DnGemmParams& p, int m, int n) {
  return p.out_layout ? ((int64_t)(n >> 5) * p.M + m) * 32 + (n & 31) : (int64_t)m * p.ldo + n;
}

template <bool H16>
__device__ __forceinline__ void store8_h16(void* base, int64_t off, const float (&v)[8]) {
  *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(base) + off) =
      make_uint4(pack_h2<H16>(v[0], v[1]), pack_h2<H16>(v[2], v[3]), pack_h2<H16>(v[4], v[5]), pack_h2<H16>(v[6], v[7]));
}

// WIDE (bf16 output, rows 16-byte aligned, BIAS / SILU / GEGLU): a lane owns 8 output columns instead of 4.
template <int EPI, bool FULL, bool H16>
__device__ __forceinline__ void wave_epilogue_wide(const DnGemmParams& p, const float* ep, int m_base, int n_base, int g, int lane,
                                                   int ncols, bool prescaled) {
  const float* bias = p.bias ? p.bias + p.bias_gstride * g : nullptr;
  char* out = reinterpret_cast<char*>(p.out) + p.out_gstride * g * 2;
  if constexpr (EPI == DN_EPI_GEGLU) {
    // packed columns come in 16-column tiles [8 value | 8 gate]: a slab of 64 carries 32 output columns; 4 lanes per row
    // (one tile = 8 outputs each), 16 rows per pass
    const int c8 = (lane & 3) * 8;
    const int sv = (lane & 3) * 16;    // slab column of this lane's value octet; its gate octet follows at + 8
    const int np = n_base + sv;        // packed row of the value octet (bias / row_bias index)
    const int n = (n_base >> 1) + c8;  // output column
    if (n >= p.N || sv >= ncols) return;
    float bv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (bias) {
      *reinterpret_cast<float4*>(bv) = *reinterpret_cast<const float4*>(bias + np);
      *reinterpret_cast<float4*>(bv + 4) = *reinterpret_cast<const float4*>(bias + np + 4);
      *reinterpret_cast<float4*>(bg) = *reinterpret_cast<const float4*>(bias + np + 8);
      *reinterpret_cast<float4*>(bg + 4) = *reinterpret_cast<const float4*>(bias + np + 12);
    }
    // beta . W^T: per sample -> fetched per row; one row for the batch -> fetched once.  Either way it enters as
    // acc * scale + rb + bias in the same order, so shared_t and per-sample t give bit-identical results.
    const bool rb_rows = p.row_ssq && p.row_bias && p.row_bias_ld != 0;
    float rsv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rsg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (p.row_ssq && p.row_bias && !rb_rows) {
      *reinterpret_cast<float4*>(rsv) = *reinterpret_cast<const float4*>(p.row_bias + np);
      *reinterpret_cast<float4*>(rsv + 4) = *reinterpret_cast<const float4*>(p.row_bias + np + 4);
      *reinterpret_cast<float4*>(rsg) = *reinterpret_cast<const float4*>(p.row_bias + np + 8);
      *reinterpret_cast<float4*>(rsg + 4) = *reinterpret_cast<const float4*>(p.row_bias + np + 12);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = j * 16 + (lane >> 2);
      const int m = m_base + row;
      if (!FULL && m >= p.M) continue;
      float v[8], gt[8], o[8];
      *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(ep + row * EP_LD + sv);
      *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(ep + row * EP_LD + sv + 4);
      *reinterpret_cast<float4*>(gt) = *reinterpret_cast<const float4*>(ep + row * EP_LD + sv + 8);
      *reinterpret_cast<float4*>(gt + 4) = *reinterpret_cast<const float4*>(ep + row * EP_LD + sv + 12);
      if (p.row_ssq) {  // split norm: scale the accumulators by the row's factor (unless the slab copy did), add beta . W^T
        const float sm = prescaled ? 1.f : ep[row * EP_LD + 64];
        float rv[8], rg[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { rv[i] = rsv[i]; rg[i] = rsg[i]; }
        if (rb_rows) {
          const float* rb = row_bias_of(p, m);
          *reinterpret_cast<float4*>(rv) = *reinterpret_cast<const float4*>(rb + np);
          *reinterpret_cast<float4*>(rv + 4) = *reinterpret_cast<const float4*>(rb + np + 4);
          *reinterpret_cast<float4*>(rg) = *reinterpret_cast<const float4*>(rb + np + 8);
          *reinterpret_cast<float4*>(rg + 4) = *reinterpret_cast<const float4*>(rb + np + 12);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[i] = __fadd_rn(__fmul_rn(v[i], sm), rv[i]); gt[i] = __fadd_rn(__fmul_rn(gt[i], sm), rg[i]); }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) { v[i] += bv[i]; gt[i] += bg[i]; o[i] = gelu_erf(gt[i]) * v[i]; }
      // K-blocked output ([N/32][M][32]): the 8 columns stay inside one 32-column block
      store8_h16<H16>(out, out_off(p, m, n), o);
      if (p.pre_out) {  // training: the pre-activation, packed columns, for the backward pass
        store8_h16<H16>(p.pre_out, (int64_t)m * p.pre_ld + np, v);
        store8_h16<H16>(p.pre_out, (int64_t)m * p.pre_ld + np + 8, gt);
      }
    }
  } else {
    const int c8 = (lane & 7) * 8;
    const int n = n_base + c8;
    if (n >= p.N || c8 >= ncols) return;
    float bv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (bias) {
      *reinterpret_cast<float4*>(bv) = *reinterpret_cast<const float4*>(bias + n);
      *reinterpret_cast<float4*>(bv + 4) = *reinterpret_cast<const float4*>(bias + n + 4);
    }
    const bool rb_rows = p.row_ssq && p.row_bias && p.row_bias_ld != 0;
    float rsv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (p.row_ssq && p.row_bias && !rb_rows) {
      *reinterpret_cast<float4*>(rsv) = *reinterpret_cast<const float4*>(p.row_bias + n);
      *reinterpret_cast<float4*>(rsv + 4) = *reinterpret_cast<const float4*>(p.row_bias + n + 4);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = j * 8 + (lane >> 3);
      const int m = m_base + row;
      if (!FULL && m >= p.M) continue;
      float v[8];
      *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(ep + row * EP_LD + c8);
      *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(ep + row * EP_LD + c8 + 4);
      if (p.row_ssq) {
        const float sm = prescaled ? 1.f : ep[row * EP_LD + 64];
        float rv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) rv[i] = rsv[i];
        if (rb_rows) {
          const float* rb = row_bias_of(p, m);
          *reinterpret_cast<float4*>(rv) = *reinterpret_cast<const float4*>(rb + n);
          *reinterpret_cast<float4*>(rv + 4) = *reinterpret_cast<const float4*>(rb + n + 4);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __fadd_rn(__fmul_rn(v[i], sm), rv[i]);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v[i] += bv[i];
        if constexpr (EPI == DN_EPI_SILU) v[i] = (p.pad_ & 64) ? fmaxf(v[i], 0.f) : silu(v[i]);  // pad_ bit 6: DN_EPI_RELU rides on this epilogue
      }
      store8_h16<H16>(out, out_off(p, m, n), v);
    }
  }
}

// pre_res (HAS_PRE, RESADD only): the residual rows of this lane, requested by the caller before its K loop in this
// function's own (row batch, lane) pattern -- v[r] = res[m_base + 4 r + (lane >> 4)][n_base + 4 (lane & 15) ..+3]; `on` false:
// not requested (A/B timing), load here.
struct ResPre { float4 v[16]; bool on = false; };
template <int EPI, int OUT_BF, bool RES_BF, bool FULL, bool HAS_PRE = false, bool SPLIT = false, bool H16 = false>  // OUT_BF: store4t's KIND (1 = the kernel's 2-byte type: bf16 or, H16, IEEE half)
__device__ __forceinline__ void wave_epilogue_impl(const DnGemmParams& p, const float* ep, int m_base, int n_base, int g, int lane,
                                                   int ncols, bool prescaled, const ResPre pre_res = ResPre()) {
  const float* bias = p.bias ? p.bias + p.bias_gstride * g : nullptr;
  char* out = reinterpret_cast<char*>(p.out) + p.out_gstride * g * (OUT_BF == 1 ? 2 : 4);
  if constexpr (EPI == DN_EPI_GEGLU) {
    // packed columns come in 16-column tiles [8 value | 8 gate]: a slab of 64 carries 32 output columns; 8 lanes per row
    // (half a tile = 4 outputs each), 8 rows per pass
    const int c4 = (lane & 7) * 4;
    const int sv = (c4 >> 3) * 16 + (c4 & 7);  // slab column of this lane's 4 values; their gates follow at + 8
    const int np = n_base + sv;                // packed row of the first value (bias / row_bias index)
    const int n = (n_base >> 1) + c4;          // output column
    if (n >= p.N || sv >= ncols) return;
    const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + np) : make_float4(0, 0, 0, 0);
    const float4 bg = bias ? *reinterpret_cast<const float4*>(bias + np + 8) : make_float4(0, 0, 0, 0);
    const bool rb_rows = p.row_ssq && p.row_bias && p.row_bias_ld != 0;  // per-sample beta . W^T: fetched per row
    float4 rsv = make_float4(0, 0, 0, 0), rsg = rsv;                      // one row for the batch: fetched once
    if (p.row_ssq && p.row_bias && !rb_rows) {
      rsv = *reinterpret_cast<const float4*>(p.row_bias + np);
      rsg = *reinterpret_cast<const float4*>(p.row_bias + np + 8);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = j * 8 + (lane >> 3);
      const int m = m_base + row;
      if (!FULL && m >= p.M) continue;
      float4 v = *reinterpret_cast<const float4*>(ep + row * EP_LD + sv);
      float4 gt = *reinterpret_cast<const float4*>(ep + row * EP_LD + sv + 8);
      if (p.row_ssq) {
        const float sm = prescaled ? 1.f : ep[row * EP_LD + 64];
        float4 rv = rsv, rg = rsg;
        if (rb_rows) {
          const float* rb = row_bias_of(p, m);
          rv = *reinterpret_cast<const float4*>(rb + np);
          rg = *reinterpret_cast<const float4*>(rb + np + 8);
        }
        v = make_float4(__fadd_rn(__fmul_rn(v.x, sm), rv.x), __fadd_rn(__fmul_rn(v.y, sm), rv.y), __fadd_rn(__fmul_rn(v.z, sm), rv.z),
                        __fadd_rn(__fmul_rn(v.w, sm), rv.w));
        gt = make_float4(__fadd_rn(__fmul_rn(gt.x, sm), rg.x), __fadd_rn(__fmul_rn(gt.y, sm), rg.y), __fadd_rn(__fmul_rn(gt.z, sm), rg.z),
                         __fadd_rn(__fmul_rn(gt.w, sm), rg.w));
      }
      v = make_float4(v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w);
      gt = make_float4(gt.x + bg.x, gt.y + bg.y, gt.z + bg.z, gt.w + bg.w);
      store4t<(OUT_BF == 1 && H16) ? 3 : OUT_BF>(out, out_off(p, m, n), gelu_erf(gt.x) * v.x, gelu_erf(gt.y) * v.y, gelu_erf(gt.z) * v.z,
                                                 gelu_erf(gt.w) * v.w);
      if (p.pre_out) {  // training: the pre-activation, packed columns, for the backward pass
        store4t<(OUT_BF == 1 && H16) ? 3 : OUT_BF>(p.pre_out, (int64_t)m * p.pre_ld + np, v.x, v.y, v.z, v.w);
        store4t<(OUT_BF == 1 && H16) ? 3 : OUT_BF>(p.pre_out, (int64_t)m * p.pre_ld + np + 8, gt.x, gt.y, gt.z, gt.w);
      }
    }
  } else {
    const int c4 = (lane & 15) * 4;
    const int n = n_base + c4;
    if (n >= p.N || c4 >= ncols) return;  // ncols < 64: only the slab's first columns carry this wave's outputs
    const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0, 0, 0, 0);
    bool rb_rows = false;
    float4 rsv = make_float4(0, 0, 0, 0);
    if constexpr (EPI == DN_EPI_BIAS || EPI == DN_EPI_SILU) {
      rb_rows = p.row_ssq && p.row_bias && p.row_bias_ld != 0;
      if (p.row_ssq && p.row_bias && !rb_rows) rsv = *reinterpret_cast<const float4*>(p.row_bias + n);
    }
    // The side inputs of a row (residual, FiLM gamma/beta, positional row) are global loads; issued inside the
    // per-row loop each would sit behind the previous row's store (the compiler must assume `out` aliases them) and
    // expose a full memory round trip per row.  So the rows go in batches of RB: all loads of a batch first, then
    // the arithmetic and the stores.
    constexpr int RB = EPI == DN_EPI_FILM_GATE ? 4 : 8;  // FiLM rows carry up to three side vectors each
    constexpr bool HAS_RES = EPI == DN_EPI_FILM_GATE || EPI == DN_EPI_RESADD;
    int b0 = 0, t0 = 0;  // (sequence, frame) of this lane's first row, advanced incrementally (no division per row)
    if constexpr (EPI == DN_EPI_FILM_GATE || EPI == DN_EPI_POSEMB) {
      const int m_first = m_base + (lane >> 4);
      b0 = m_first / p.T;
      t0 = m_first - b0 * p.T;
    }
    const char* resb = nullptr;
    if constexpr (HAS_RES) resb = reinterpret_cast<const char*>(p.res) + p.res_gstride * g * (RES_BF ? 2 : 4);
    const float* gbb = nullptr;
    if constexpr (EPI == DN_EPI_FILM_GATE) gbb = p.gamma_beta ? p.gamma_beta + p.gb_gstride * g + n : nullptr;
    const bool gb_shared = p.gb_ld == 0;
    float4 split_gamma = make_float4(1, 1, 1, 1);  // split norm: the gamma row when one row serves the whole batch
    if constexpr (EPI == DN_EPI_RESADD || EPI == DN_EPI_POSEMB) {
      if (p.norm_split) {
        if (p.norm_gb && !p.norm_gb_ld) split_gamma = *reinterpret_cast<const float4*>(p.norm_gb + n);
        else if (!p.norm_gb && p.norm_gamma) split_gamma = *reinterpret_cast<const float4*>(p.norm_gamma + n);
      }
    }
    float4 gas = make_float4(1, 1, 1, 1), bes = make_float4(0, 0, 0, 0);
    if constexpr (EPI == DN_EPI_FILM_GATE)
      if (gbb && gb_shared) {  // one conditioning row for the whole batch (sampling): load it once per lane
        gas = *reinterpret_cast<const float4*>(gbb);
        bes = *reinterpret_cast<const float4*>(gbb + p.gb_half);
      }
#pragma unroll
    for (int jb = 0; jb < 16; jb += RB) {
      float4 rv[RB], ga[RB], be[RB];
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int m = m_base + (jb + i) * 4 + (lane >> 4);
        rv[i] = make_float4(0, 0, 0, 0);
        ga[i] = gas;
        be[i] = bes;
        if (!FULL && m >= p.M) continue;
        if constexpr (HAS_RES) {
          if constexpr (HAS_PRE) {
            if (pre_res.on) rv[i] = pre_res.v[jb + i];
            else rv[i] = load4t<RES_BF ? (H16 ? 3 : 1) : 0>(resb, (int64_t)m * p.ldr + n);
          } else {
            rv[i] = load4t<RES_BF ? (H16 ? 3 : 1) : 0>(resb, (int64_t)m * p.ldr + n);
          }
        }
        if constexpr (EPI == DN_EPI_FILM_GATE) {
          if (gbb && !gb_shared) {
            const float* gr = gbb + (int64_t)b0 * p.gb_ld;
            ga[i] = *reinterpret_cast<const float4*>(gr);
            be[i] = *reinterpret_cast<const float4*>(gr + p.gb_half);
          }
        }
        if constexpr (EPI == DN_EPI_POSEMB) {
          const int pos = t0 < p.lengths[b0] ? t0 + 1 : 0;
          rv[i] = *reinterpret_cast<const float4*>(p.pos_table + (int64_t)pos * p.pos_ld + n);
        }
        if constexpr (EPI == DN_EPI_FILM_GATE || EPI == DN_EPI_POSEMB) {
          if (p.T >= 4) {  // next row of this lane is 4 frames on: at most one sequence boundary
            t0 += 4;
            if (t0 >= p.T) {
              t0 -= p.T;
              ++b0;
            }
          } else {
            b0 = (m + 4) / p.T;
            t0 = (m + 4) - b0 * p.T;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int row = (jb + i) * 4 + (lane >> 4);
        const int m = m_base + row;
        if (!FULL && m >= p.M) continue;
        float4 a4 = *reinterpret_cast<const float4*>(ep + row * EP_LD + c4);
        if constexpr (EPI == DN_EPI_BIAS || EPI == DN_EPI_SILU) {
          if (p.row_ssq) {  // split norm, consumer side
            const float sm = prescaled ? 1.f : ep[row * EP_LD + 64];
            float4 rb4 = rsv;
            if (rb_rows) rb4 = *reinterpret_cast<const float4*>(row_bias_of(p, m) + n);
            a4 = make_float4(__fadd_rn(__fmul_rn(a4.x, sm), rb4.x), __fadd_rn(__fmul_rn(a4.y, sm), rb4.y),
                             __fadd_rn(__fmul_rn(a4.z, sm), rb4.z), __fadd_rn(__fmul_rn(a4.w, sm), rb4.w));
          }
        }
        float v0 = a4.x + bv.x, v1 = a4.y + bv.y, v2 = a4.z + bv.z, v3 = a4.w + bv.w;
        if constexpr (EPI == DN_EPI_SILU) {
          if (p.pad_ & 64) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }  // DN_EPI_RELU
          else { v0 = silu(v0); v1 = silu(v1); v2 = silu(v2); v3 = silu(v3); }
        } else if constexpr (EPI == DN_EPI_FILM_GATE) {
          if (gbb) {
            v0 = v0 * ga[i].x + be[i].x; v1 = v1 * ga[i].y + be[i].y; v2 = v2 * ga[i].z + be[i].z; v3 = v3 * ga[i].w + be[i].w;
          }
          v0 = tanh_sigmoid_gate(v0) + rv[i].x; v1 = tanh_sigmoid_gate(v1) + rv[i].y;
          v2 = tanh_sigmoid_gate(v2) + rv[i].z; v3 = tanh_sigmoid_gate(v3) + rv[i].w;
        } else if constexpr (EPI == DN_EPI_RESADD || EPI == DN_EPI_POSEMB) {
          v0 += rv[i].x; v1 += rv[i].y; v2 += rv[i].z; v3 += rv[i].w;
        }
        store4t<(OUT_BF == 1 && H16) ? 3 : OUT_BF>(out, out_off(p, m, n), v0, v1, v2, v3);
        if constexpr (EPI == DN_EPI_RESADD || EPI == DN_EPI_POSEMB) {
          if (p.norm_split) {  // split norm, producer side: row * gamma for the consuming contraction + this slab's sum of squares
            float4 ga = split_gamma;
            if (p.norm_gb && p.norm_gb_ld) ga = *reinterpret_cast<const float4*>(p.norm_gb + (int64_t)(m / p.T) * p.norm_gb_ld + n);
            // norm_split == 2: K-blocked [norm_ld/32][M][32] for a consumer that stages whole cache lines (a lane's 4 columns stay in one block)
            const int64_t noff = p.norm_split == 2 ? ((int64_t)(n >> 5) * p.M + m) * 32 + (n & 31) : (int64_t)m * p.norm_ld + n;
            if (p.norm_dtype == (SPLIT ? DN_BF16X3 : H16 ? DN_F16 : DN_BF16)) store4t<SPLIT ? 2 : H16 ? 3 : 1>(p.norm_out, noff, v0 * ga.x, v1 * ga.y, v2 * ga.z, v3 * ga.w);
            else store4t<0>(p.norm_out, noff, v0 * ga.x, v1 * ga.y, v2 * ga.z, v3 * ga.w);
            float q = v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;  // the row's 16 lanes are lanes (lane & ~15) .. +15
            q = row16_sum(q);
            if ((lane & 15) == 0) p.norm_ssq[(int64_t)m * p.norm_ssq_ld + (n_base >> 6)] = q;
          }
        }
      }
    }
  }
}

// SPLIT: the kernel runs split operands (DN_BF16X3): its outputs are split rows or fp32, never plain bf16 -- and a kernel of the
// other arithmetics never writes split rows -- so each kernel carries the epilogue for two storage kinds only (a third set of
// inlined epilogues made the FiLM kernels large enough for the compiler to keep the parameter block in scratch).
template <int EPI, bool SPLIT = false, bool H16 = false>
__device__ __forceinline__ void wave_epilogue(const DnGemmParams& p, const float* ep, int m_base, int n_base, int g, int lane,
                                              int ncols = 64, float row_scale = -1.f, const ResPre pre_res = ResPre()) {
  if constexpr ((DN_GEMM_ABL & 16) != 0) return;  // diagnostic build: no epilogue (LDS reads, math and stores skipped)
  const bool full = m_base + 64 <= p.M;          // wave-uniform: slab entirely inside M
  const bool obf = !SPLIT && p.out_dtype == (H16 ? DN_F16 : DN_BF16);   // kernel arguments: uniform (the kernel's own 2-byte type)
  const bool osp = SPLIT && p.out_dtype == DN_BF16X3;  // split rows (the residual stream and dense fp32 outputs never are)
  constexpr bool RESADD = EPI == DN_EPI_RESADD;  // the residual stream is always fp32 (in and out)
  const bool rbf = EPI == DN_EPI_FILM_GATE && p.res_dtype == (H16 ? DN_F16 : DN_BF16);
  const bool prescaled = row_scale == ROW_PRESCALED;  // the slab already holds acc * sqrt(D)/|row|
  if constexpr (EPI == DN_EPI_BIAS || EPI == DN_EPI_SILU || EPI == DN_EPI_GEGLU) {
    if (p.row_ssq && row_scale != ROW_PRESCALED) {
      if (row_scale >= 0.f) const_cast<float*>(ep)[lane * EP_LD + 64] = row_scale;  // requested at kernel start
      else park_row_scales(p, const_cast<float*>(ep), m_base, lane);
    }
  }
#define DN_EP(O, R, F) wave_epilogue_impl<EPI, O, R, F, false, SPLIT, H16>(p, ep, m_base, n_base, g, lane, ncols, prescaled)
  if constexpr (RESADD) {
    if (full) wave_epilogue_impl<EPI, 0, false, true, true, SPLIT, H16>(p, ep, m_base, n_base, g, lane, ncols, prescaled, pre_res);
    else wave_epilogue_impl<EPI, 0, false, false, true, SPLIT, H16>(p, ep, m_base, n_base, g, lane, ncols, prescaled, pre_res);
  } else if constexpr (EPI == DN_EPI_FILM_GATE) {
    if constexpr (SPLIT) {  // (split mode keeps the residual branch fp32)
      if (osp) { if (full) DN_EP(2, false, true); else DN_EP(2, false, false); }
      else { if (full) DN_EP(0, false, true); else DN_EP(0, false, false); }
    } else {
      if (obf && rbf) { if (full) DN_EP(1, true, true); else DN_EP(1, true, false); }
      else if (!obf && !rbf) { if (full) DN_EP(0, false, true); else DN_EP(0, false, false); }
      else if (obf) DN_EP(1, false, false);
      else DN_EP(0, true, false);
    }
  } else {
    if constexpr (EPI == DN_EPI_BIAS || EPI == DN_EPI_SILU || EPI == DN_EPI_GEGLU) {
      // 16-byte stores need 8-column granularity and 16-byte aligned rows (uniform: kernel arguments)
      const bool wide = !SPLIT && !(p.pad_ & 16) && obf && (p.N & 7) == 0 && (p.ldo & 7) == 0 && (ncols & 7) == 0 && (p.out_gstride & 7) == 0 &&
                        (reinterpret_cast<uintptr_t>(p.out) & 15) == 0;
      if (wide) {
        if (full) wave_epilogue_wide<EPI, true, H16>(p, ep, m_base, n_base, g, lane, ncols, prescaled);
        else wave_epilogue_wide<EPI, false, H16>(p, ep, m_base, n_base, g, lane, ncols, prescaled);
        return;
      }
    }
    if constexpr (SPLIT) {
      if (osp) { if (full) DN_EP(2, false, true); else DN_EP(2, false, false); }
      else { if (full) DN_EP(0, false, true); else DN_EP(0, false, false); }
    } else {
      if (obf) { if (full) DN_EP(1, false, true); else DN_EP(1, false, false); }
      else { if (full) DN_EP(0, false, true); else DN_EP(0, false, false); }
    }
  }
#undef DN_EP
}

// Tile geometry: BM activation rows x 128 weight rows, BM/32 waves (each 64 x 64), STAGES-deep LDS ring.
//   BM = 256: 8 waves, 48 KiB per stage, 3 stages (144 KiB, one workgroup per CU): two K-tiles in flight
//             while the third is consumed -- the K loop is bound by L2/HBM latency, not bandwidth, and this
//             is what keeps ~96 KiB per CU in flight.
//   BM = 128: 4 waves, 32 KiB per stage, 2 stages (two workgroups per CU) for problems with few tiles.
template <typename E, int EPI, int BM, int STAGES>
__global__ __launch_bounds__(BM * 2, 1) void conv_gemm_kernel(const DnGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if constexpr (std::is_same<E, F16>::value) f16_saturate();  // stores of a value beyond 65504 clamp instead of becoming inf
  constexpr int ES = Elem<E>::bytes;
  constexpr int KT = ROWB / ES;                     // K elements per K-tile
  constexpr int NWAVES = BM / 32;
  constexpr int WPW = 16 / NWAVES;                  // weight pieces (8 rows each) staged per wave
  constexpr int PER_STAGE = 4 + WPW;                // LDS-DMA instructions per wave per stage
  constexpr int STAGE_BYTES = W_TILE_BYTES + BM * ROWB;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int g = blockIdx.y;

  // XCD-aware tile order: blocks sharing blockIdx % 8 share an L2; give each XCD a contiguous run of
  // logical tiles with n fastest so the tiles that re-read one A row-panel sit behind one L2.
  const int n_tiles_n = (p.N * (EPI == DN_EPI_GEGLU ? 2 : 1) + BN - 1) / BN;
  int logical;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int mt_, nt_;
  tile_coords(logical, n_tiles_n, (p.M + BM - 1) / BM, (p.pad_ >> 24) & 0xff, mt_, nt_);
  const int m0 = mt_ * BM;
  const int n0 = nt_ * BN;  // packed weight row of the tile

  // ---- staging geometry: wave w stages A rows [32w, 32w+32) and W rows [8*WPW*w, ...), 8 rows per piece ----
  const int srow = lane >> 3;                 // row within the 8-row piece
  const int schunk = (lane & 7) ^ srow;       // swizzled 16-byte chunk this lane fetches
  int a_row[4], a_t[4];                       // (clamped) row m of A and its frame index within the sequence
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + wave * 32 + i * 8 + srow;
    m = m < p.M ? m : p.M - 1;
    a_row[i] = m;
    a_t[i] = m % p.T;
  }
  const int ktiles_per_term = p.K / KT;
  const int nkt = p.n_terms * ktiles_per_term;
  const char* zero_src = reinterpret_cast<const char*>(g_zero_page) + schunk * 16;

  // Residual-closing contractions (RESADD; N = 512: one round of tiles, every workgroup a bare latency chain of staging, a short
  // K loop and a load-add-store epilogue): the wave's residual rows are requested here, in the epilogue's own access pattern,
  // so their round trip runs under the staging prologue and the K loop instead of in front of the stores.  Plain loads: they
  // are older than every LDS-DMA piece (in-order return: the K loop's counted waits cover them) and first used after it.
  ResPre pre_res;
  pre_res.on = false;
  if constexpr (EPI == DN_EPI_RESADD && BM == 128) {  // (the 8-wave tile has no 64 registers to spare)
    pre_res.on = !(p.pad_ & 32);  // pad_ bit 5: A/B timing without the prefetch
    const int n_ = n0 + wn * 64 + (lane & 15) * 4;
    const char* resb = reinterpret_cast<const char*>(p.res) + p.res_gstride * g * 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m_ = m0 + wm * 64 + r * 4 + (lane >> 4);
      pre_res.v[r] = make_float4(0, 0, 0, 0);
      if (pre_res.on && m_ < p.M && n_ < p.N) pre_res.v[r] = load4t<0>(resb, (int64_t)m_ * p.ldr + n_);
    }
  }

  // Per-lane source pointers of the term being staged; they advance by one K-tile (128 B) per stage and
  // are rebuilt only at a term boundary, so the steady-state loop carries a few pointer adds, no multiplies.
  const char* a_ptr[4];
  const char* w_ptr[WPW];
  int a_inc[4];
  int s_term = 0, s_kk = 0;
  auto setup_term = [&](int term) {
    const DnGemmTerm& tm = p.terms[term];
    const int shift = tm.shift_by_group ? tm.shift * (1 << g) : tm.shift;
    const char* A = reinterpret_cast<const char*>(tm.A) + (tm.a_gstride * g) * ES + schunk * 16;
    const char* W = reinterpret_cast<const char*>(tm.W) + (tm.w_gstride * g) * ES + schunk * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool valid = shift_valid(a_t[i], shift, p.T);  // frames before the sequence start read the zero page
      a_ptr[i] = valid ? A + (int64_t)(a_row[i] - shift) * tm.lda * ES : zero_src;
      a_inc[i] = valid ? ROWB : 0;
    }
#pragma unroll
    for (int i = 0; i < WPW; ++i) w_ptr[i] = W + (int64_t)(n0 + (wave * WPW + i) * 8 + srow) * (tm.ldw ? tm.ldw : p.K) * ES;
  };
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lptr_t)smem);
  auto stage = [&](int slot) {
    const uint32_t sbase = lds_base + slot * STAGE_BYTES;
    const uint32_t wbase = sbase + wave * (WPW * 1024);
    const uint32_t abase = sbase + W_TILE_BYTES + wave * 4096;
#pragma unroll
    for (int i = 0; i < WPW; ++i) glds16(w_ptr[i], wbase + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(a_ptr[i], abase + i * 1024);
    if (++s_kk == ktiles_per_term) {
      s_kk = 0;
      if (++s_term < p.n_terms) setup_term(s_term);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) a_ptr[i] += a_inc[i];
#pragma unroll
      for (int i = 0; i < WPW; ++i) w_ptr[i] += ROWB;
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read addresses: row (l & 15) of a 16-row sub-tile, chunk ks*4 + (l >> 4), swizzled by row & 7
  const int frow = lane & 15, fq = lane >> 4;
  const int w_rd = (wn * 64 + frow) * ROWB;
  const int a_rd = W_TILE_BYTES + (wm * 64 + frow) * ROWB;
  const int sw = frow & 7;

  uint4 wf[2][4], af[2][4];  // fragments of one K-tile: [k-step][16-row sub-tile]
  auto load_frags = [&](int slot) {
    const char* sb = smem + slot * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + fq) ^ sw) << 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[ks][i] = *reinterpret_cast<const uint4*>(sb + w_rd + i * 16 * ROWB + coff);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[ks][i] = *reinterpret_cast<const uint4*>(sb + a_rd + i * 16 * ROWB + coff);
    }
  };
  auto mma_all = [&]() {
    if constexpr (std::is_same<E, BF16X3>::value) {
      // split rows: the two k-steps of the 128-byte K-tile are the halves of 32 elements -- activations hi then lo, weights lo
      // then hi (include/diffnorm_hip.h).  Three bf16 MFMAs per product, small terms first; the 16 accumulators of a pass
      // separate the dependent MFMAs of one accumulator.
#pragma unroll
      for (int pass = 0; pass < 3; ++pass)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
            mma_kstep<BF16>(acc[nt][mt], wf[pass == 0 ? 0 : 1][nt], af[pass == 1 ? 1 : 0][mt]);  // w_lo.a_hi, w_hi.a_lo, w_hi.a_hi
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) mma_kstep<E>(acc[nt][mt], wf[ks][nt], af[ks][mt]);
    }
  };

  const RowSsqReq rs_req = row_scale_request<EPI>(p, m0 + wm * 64, lane);
  setup_term(0);
  if constexpr (STAGES == 3) {
    // Staggered two-group schedule (8 waves = group 0: waves 0-3, group 1: waves 4-7; waves w and w+4
    // share a SIMD).  Every K-tile is two barrier-separated segments per wave: L = issue the LDS-DMA of
    // tile k+2 and pull tile k's fragments LDS -> registers, C = 32 MFMAs from registers.  Group 1 runs one
    // segment behind group 0, so on every SIMD one wave is in its matrix segment while its partner is in
    // its memory segment.  Segment s = 2k / 2k+1 is (L_k, C_k) for group 0 and s = 2k+1 / 2k+2 for group 1.
    //   RAW: tile k+1 must be in LDS before segment 2k+2; every wave retires its own pieces with a counted
    //        vmcnt ahead of the barrier that ends odd segment 2k+1 (only tile k+2's 6 pieces stay in flight).
    //   WAR: tile k+2 overwrites the slot of tile k-1, whose last reads (group 1, segment 2k-1) finished
    //        before the barrier that opens segment 2k, the earliest segment that issues tile k+2.
    const bool late = __builtin_amdgcn_readfirstlane(wave) >= NWAVES / 2;
    stage(0);
    if (nkt > 1) stage(1);
    if (nkt > 1) pipe_sync<PER_STAGE>(); else pipe_sync<0>();  // tile 0 landed
    __builtin_amdgcn_sched_barrier(0);
    if (late) pipe_sync<63>();  // the stagger: group 1 sits out segment 0 (vmcnt(63) = no wait)
    // The K loop is specialised on the wave's group and has no per-iteration conditions (a uniform branch in the wave
    // that is feeding the MFMA pipe idles the pipe): tiles that still stage (kt + 2 < nkt) run in the main loop, the last
    // two drain.
    auto ktile = [&](auto late_c, auto stage_c, int slot, int fill) {
      constexpr bool LATE = decltype(late_c)::value, STAGE = decltype(stage_c)::value;
      // ---- L segment
      if constexpr (!(DN_GEMM_ABL & 4)) load_frags(slot);  // LDS reads first: they drain while the TA chews the DMA addresses
      if constexpr (STAGE && !(DN_GEMM_ABL & 1)) stage(fill);
      if constexpr (LATE) pipe_sync<STAGE ? PER_STAGE : 0>(); else pipe_sync<63>();  // an odd segment for group 1
      __builtin_amdgcn_sched_barrier(0);
      // ---- C segment
      if constexpr (!(DN_GEMM_ABL & 8)) __builtin_amdgcn_s_setprio(1);
      if constexpr (!(DN_GEMM_ABL & 2)) mma_all();
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (LATE) pipe_sync<63>(); else pipe_sync<STAGE ? PER_STAGE : 0>();  // odd segment for group 0
      __builtin_amdgcn_sched_barrier(0);
    };
    auto run = [&](auto late_c) {
      int slot = 0, fill = 2, kt = 0;
      auto adv = [&]() { slot = slot == 2 ? 0 : slot + 1; fill = fill == 2 ? 0 : fill + 1; };
      for (; kt + 2 < nkt; ++kt) { ktile(late_c, std::true_type{}, slot, fill); adv(); }
      for (; kt < nkt; ++kt) { ktile(late_c, std::false_type{}, slot, fill); adv(); }
    };
    if (late) run(std::true_type{}); else run(std::false_type{});
    if (!late) pipe_sync<63>();  // group 0 matches group 1's extra barrier
  } else {
    // Two-stage loop for the 4-wave tile (two workgroups per CU overlap each other): one barrier per K-tile.
    stage(0);
    int slot = 0;
    for (int kt = 0; kt < nkt; ++kt) {
      pipe_sync<0>();  // tile kt landed for everyone, tile kt-1 fully consumed
      if (kt + 1 < nkt) stage(slot ^ 1);
      load_frags(slot);
      mma_all();
      slot ^= 1;
    }
  }

  // ------------------------------------------------------------------ epilogue
  // The accumulators hold, per lane, 4 consecutive columns of 16 different rows; stored from there a wave
  // instruction would touch 16 rows x 32 bytes.  Each wave therefore transposes its 64 x 64 fp32 tile through a
  // private LDS slab ([64][68] floats, the ring is dead by now) and runs the epilogue row-wise: 16 lanes cover the
  // 64 columns of one row, so bias / FiLM / residual loads and the output stores are whole 128-256 B row segments.
  if constexpr (STAGES != 3) __syncthreads();  // the 2-stage loop ends without a barrier: ring reads must be over
  float* ep = reinterpret_cast<float*>(smem) + wave * (64 * EP_LD);
  const float row_scale = row_scale_finish(p, rs_req);
  float sc4[4];  // split RMSNorm, consumer side: the row factor is applied while the accumulators go to the slab
  slab_row_scales(row_scale, lane, sc4);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      *reinterpret_cast<f32x4*>(ep + (mt * 16 + frow) * EP_LD + nt * 16 + fq * 4) = scaled(acc[nt][mt], sc4[mt]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave, LDS is in-order: writes precede the reads below

  if constexpr (EPI == DN_EPI_RESADD)
    wave_epilogue<EPI, std::is_same<E, BF16X3>::value, std::is_same<E, F16>::value>(p, ep, m0 + wm * 64, n0 + wn * 64, g, lane, 64, row_scale >= 0.f ? ROW_PRESCALED : row_scale, pre_res);
  else
    wave_epilogue<EPI, std::is_same<E, BF16X3>::value, std::is_same<E, F16>::value>(p, ep, m0 + wm * 64, n0 + wn * 64, g, lane, 64, row_scale >= 0.f ? ROW_PRESCALED : row_scale);
}

// ------------------------------------------------------------------------------------------ 256 x 256 tile
// The large-problem variant.  At 256 x 128 the K loop is pinned by two per-CU paths that cannot be made faster, only
// relieved: the L2 -> LDS DMA path (~60 B/clk: 48 KiB per K-tile ~ 800 cycles) and the LDS fragment reads, each about as
// long as the 1024 MFMA cycles they feed (ablation in DESIGN.md).  Doubling the weight tile halves both per MFMA:
//   tile 256 (activation rows) x 256 (weight rows), 8 waves, each 64 (m) x 128 (n): 32 MFMAs per 32-deep K-tile from
//   12 fragment reads (16 -> 12 per 32 MFMAs) and 4 DMA pieces (6 -> 4);
//   K-tiles are 64 bytes per row (32 bf16 / 16 f32) so a 4-stage ring is 128 KiB: three K-tiles in flight;
//   64-byte rows put 4 rows in a bank row: the conflict-free swizzle is chunk ^= 2 * ((row >> 3) & 1).
// Schedule: the same staggered two-group L / C segments as above, one K-tile per segment pair.
constexpr int ROWB2 = 64;

// TAPS: the terms are the taps of ONE causal conv (launch_big checks: same activation tensor and layout, shifts and weight
// addresses in arithmetic progression) and run innermost in K -- K-tile n is tap n % n_terms of K-chunk n / n_terms, the order of
// the 256 x 352 tile (same 32-deep K-tiles: bit-identical to it), so the taps of a chunk read (nearly) the same activation rows
// back to back and the XCD's L2 serves all but one of them: the dilated WaveNet conv's activation panels cross the fabric once
// instead of three times (measured 516 MB of reads per launch at [32,512], 402 MB of them the panels).
// BNB = 192: the same kernel on a 256 x 192 tile (waves 64 x 96) for widths that quantise badly on 256 columns -- N = 1408 is 5.5
// column tiles of 256 (384 workgroups = 1.5 rounds of the 256 CUs) but 7.33 of 192 (512 workgroups = 2 full rounds, 8 % of
// them padding).  The weight tile's 12 staging pieces go 2 per wave to waves 0-3 and 1 per wave to waves 4-7, so the two wave
// groups count different numbers of DMA pieces per stage.  Epilogues whose bookkeeping assumes 64-aligned wave columns (the
// split norm's producer) do not run on it.
// HALO (TAPS, three taps with shifts 2d, d, 0): as on the 256 x 352 tile, the taps share one staged copy of the tile's rows plus
// the H = 16 * ceil(2d / 16) rows in front of them (decided per tile and, with shifts scaled by the group, per group: 2d <= 128).
template <typename E, int EPI, bool TAPS, int BNB = 256, bool HALO = false>
__global__ __launch_bounds__(512, 1) void conv_gemm_big_kernel(const DnGemmParams p) {
  static_assert(!HALO || (TAPS && BNB == 256 && !std::is_same<E, BF16X3>::value), "shared staging: tap-inner order on the 256 x 256 tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if constexpr (std::is_same<E, F16>::value) f16_saturate();  // stores of a value beyond 65504 clamp instead of becoming inf
  constexpr int ES = Elem<E>::bytes;
  constexpr int KT = ROWB2 / ES;
  static_assert(BNB == 256 || BNB == 192, "tile width");
  constexpr int BMB = 256, STAGES = 4;
  constexpr int TILE = BNB * ROWB2;               // the weight tile (16 KiB at 256 columns); the row tile follows it
  constexpr int STAGE_BYTES = TILE + 256 * ROWB2;
  constexpr int PER_STAGE = 4;                    // DMA pieces per wave per stage (2 W + 2 A, 16 rows each) ...
  constexpr int PER_LATE = BNB == 256 ? 4 : 3;    // ... waves 4-7 of the 192-column tile: 1 W + 2 A
  constexpr int NTW = BNB / 32;                   // 16-column accumulator tiles per wave

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int g = blockIdx.y;
  const int np_total = p.N * (EPI == DN_EPI_GEGLU ? 2 : 1);   // packed weight rows that carry output
  const int n_tiles_n = (np_total + BNB - 1) / BNB;
  int logical;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int mt_, nt_;
  tile_coords(logical, n_tiles_n, (p.M + BMB - 1) / BMB, (p.pad_ >> 24) & 0xff, mt_, nt_);
  const int m0 = mt_ * BMB;
  const int n0 = nt_ * BNB;
  const int w_rows = (np_total + 127) / 128 * 128;  // rows the packed weight really has: clamp the ragged last tile

  // ---- staging: wave w stages rows [32w, 32w+32) of both tiles, 16 rows x 64 B per piece
  const int srow = lane >> 2;
  const int schunk = (lane & 3) ^ (((lane >> 5) & 1) << 1);
  int a_row[2], a_t[2], w_row[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = m0 + wave * 32 + i * 16 + srow;
    m = m < p.M ? m : p.M - 1;
    a_row[i] = m;
    a_t[i] = m % p.T;
    int n = n0 + (BNB == 256 || wave < 4 ? wave * 32 + i * 16 : 128 + (wave - 4) * 16) + srow;  // (192: waves 4-7 stage one piece, i = 0)
    w_row[i] = n < w_rows ? n : w_rows - 1;
  }
  const int ktiles_per_term = p.K / KT;
  const int nkt = p.n_terms * ktiles_per_term;
  const char* zero_src = reinterpret_cast<const char*>(g_zero_page) + schunk * 16;
  const char* a_ptr[2];
  const char* w_ptr[2];
  int a_inc[2], w_inc = ROWB2;
  int s_term = 0, s_kk = 0;
  auto setup_term = [&](int term) {
    const DnGemmTerm& tm = p.terms[term];
    const int shift = tm.shift_by_group ? tm.shift * (1 << g) : tm.shift;
    const char* A = reinterpret_cast<const char*>(tm.A) + (tm.a_gstride * g) * ES + schunk * 16;
    const char* W = reinterpret_cast<const char*>(tm.W) + (tm.w_gstride * g) * ES + schunk * 16;
    // K-blocked operands ([K/32][rows][32], DN_LAYOUT_*): rows are 64 bytes apart, K-tiles a whole block apart
    const bool a_kb = tm.layout & DN_LAYOUT_A_KBLOCKED, w_kb = tm.layout & DN_LAYOUT_W_KBLOCKED;
    const int64_t a_rowb = a_kb ? ROWB2 : (int64_t)tm.lda * ES, w_rowb = w_kb ? ROWB2 : (int64_t)(tm.ldw ? tm.ldw : p.K) * ES;
    const int a_step = a_kb ? p.M * ROWB2 : ROWB2;
    w_inc = w_kb ? w_rows * ROWB2 : ROWB2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool valid = shift_valid(a_t[i], shift, p.T);
      a_ptr[i] = valid ? A + (int64_t)(a_row[i] - shift) * a_rowb : zero_src;
      a_inc[i] = valid ? a_step : 0;
      w_ptr[i] = W + (int64_t)w_row[i] * w_rowb;
    }
  };
  // tap-inner: the per-tap uniforms are affine in the tap index and carried as running scalars (a_ptr / w_ptr then hold the
  // UNSHIFTED row and tap 0's weight row of the current K-chunk, a_t the frame index that decides the zero page per tap)
  int tap_shift = 0, tap_shift0 = 0, tap_sstep = 0;
  int64_t tap_delta = 0, tap_delta0 = 0, tap_dstep = 0, tap_woff = 0, tap_wstride = 0;
  int tap_ainc = ROWB2;
  [[maybe_unused]] const char* halo_ptr = nullptr;  // HALO: this lane's row of the rows in front of the tile, or the zero page
  [[maybe_unused]] int halo_inc = 0, halo_rows = 0, a_rdh[3] = {0, 0, 0};  // a_rdh: per tap, this lane's fragment offset inside an activation slot
  [[maybe_unused]] uint64_t halo_mask = 0;
  auto setup_taps = [&]() {
    const DnGemmTerm& t0 = p.terms[0];
    const DnGemmTerm& t1 = p.terms[1];
    const int sh0 = t0.shift_by_group ? (t0.shift << g) : t0.shift, sh1 = t1.shift_by_group ? (t1.shift << g) : t1.shift;
    const bool a_kb = t0.layout & DN_LAYOUT_A_KBLOCKED, w_kb = t0.layout & DN_LAYOUT_W_KBLOCKED;
    const int64_t a_rowb = a_kb ? ROWB2 : (int64_t)t0.lda * ES, w_rowb = w_kb ? ROWB2 : (int64_t)(t0.ldw ? t0.ldw : p.K) * ES;
    tap_shift0 = sh0; tap_sstep = sh0 - sh1;
    tap_delta0 = sh0 * a_rowb; tap_dstep = tap_sstep * a_rowb;
    tap_wstride = (int64_t)((intptr_t)t1.W - (intptr_t)t0.W);
    tap_shift = tap_shift0; tap_delta = tap_delta0; tap_woff = 0;
    tap_ainc = a_kb ? p.M * ROWB2 : ROWB2;
    w_inc = w_kb ? w_rows * ROWB2 : ROWB2;
    const char* A = reinterpret_cast<const char*>(t0.A) + (t0.a_gstride * g) * ES + schunk * 16;
    const char* W = reinterpret_cast<const char*>(t0.W) + (t0.w_gstride * g) * ES + schunk * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a_ptr[i] = A + (int64_t)a_row[i] * a_rowb;
      w_ptr[i] = W + (int64_t)w_row[i] * w_rowb;
    }
    if constexpr (HALO) {  // wave w stages halo rows [w H/8, (w + 1) H/8), four lanes per row
      halo_rows = (2 * sh1 + 15) / 16 * 16;
      const int per_wave = halo_rows >> 3;
      const int R = wave * per_wave + (lane >> 2);           // LDS row of the activation slot
      const int hchunk = (lane & 3) ^ (((R >> 3) & 1) << 1);
      const bool before = m0 % p.T + R - halo_rows < 0;      // a frame in front of the tile's sequence: zeros
      const char* A0 = reinterpret_cast<const char*>(t0.A) + (t0.a_gstride * g) * ES + hchunk * 16;
      halo_ptr = before ? reinterpret_cast<const char*>(g_zero_page) + hchunk * 16 : A0 + (int64_t)(m0 + R - halo_rows) * a_rowb;
      halo_inc = before ? 0 : tap_ainc;
      halo_mask = per_wave >= 16 ? ~0ull : (1ull << (4 * per_wave)) - 1;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int row = halo_rows + wm * 64 + (lane & 15) - (2 - t) * sh1;
        a_rdh[t] = row * ROWB2 + (((lane >> 4) ^ (((row >> 3) & 1) << 1)) << 4);
      }
    }
  };
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lptr_t)smem);
  auto stage = [&](int slot) {
    const bool one_w = BNB != 256 && wave >= 4;  // uniform
    const uint32_t wbase = lds_base + slot * STAGE_BYTES + (one_w ? 8192 + (wave - 4) * 1024 : wave * 2048);
    const uint32_t abase = lds_base + slot * STAGE_BYTES + TILE + wave * 2048;
    if constexpr (TAPS) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if (i == 0 || !one_w) glds16(w_ptr[i] + tap_woff, wbase + i * 1024);
#pragma unroll
      for (int i = 0; i < 2; ++i) glds16(a_t[i] >= tap_shift ? a_ptr[i] - tap_delta : zero_src, abase + i * 1024);
      const bool wrap = s_term + 1 == p.n_terms;  // uniform selects, no branch
      s_term = wrap ? 0 : s_term + 1;
      tap_shift = wrap ? tap_shift0 : tap_shift - tap_sstep;
      tap_delta = wrap ? tap_delta0 : tap_delta - tap_dstep;
      tap_woff = wrap ? 0 : tap_woff + tap_wstride;
      const int ai = wrap ? tap_ainc : 0, wi = wrap ? w_inc : 0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a_ptr[i] += ai;
        w_ptr[i] += wi;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (i == 0 || !one_w) glds16(w_ptr[i], wbase + i * 1024);
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(a_ptr[i], abase + i * 1024);
    if (++s_kk == ktiles_per_term) {
      s_kk = 0;
      if (++s_term < p.n_terms) setup_term(s_term);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a_ptr[i] += a_inc[i];
        w_ptr[i] += w_inc;
      }
    }
  };

  f32x4 acc[NTW][4];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment reads: row (l & 15) of a 16-row sub-tile, 16-byte chunk (l >> 4) of the 64-byte row, swizzled
  const int frow = lane & 15, fq = lane >> 4;
  const int coff = (fq ^ (((frow >> 3) & 1) << 1)) << 4;
  const int w_rd = (wn * (BNB / 2) + frow) * ROWB2 + coff;
  const int a_rd = TILE + (wm * 64 + frow) * ROWB2 + coff;
  // Split operands (DN_BF16X3): the 64-byte K-tiles of a row alternate between the two halves of a 32-element group -- even
  // K-tiles hold the weights' lo and the activations' hi half, odd ones the weights' hi and the activations' lo half (that is
  // why the two operand kinds store their halves in opposite orders).  Even tile: acc += w_lo . a_hi, and a_hi stays in
  // registers (`ah`); odd tile: acc += w_hi . a_lo + w_hi . a_hi.  Per pair: 96 bf16 MFMAs for 32 elements of K, the same
  // staging and the same 12 fragment reads per K-tile as plain bf16; the uneven matrix segments (32 / 64 MFMAs) of the two
  // wave groups pair up, one group's long segment running beside the other's short one.
  constexpr bool X3 = std::is_same<E, BF16X3>::value;
  using ME = typename std::conditional<X3, BF16, E>::type;  // the MFMA's operand type
  uint4 wf[NTW], af[4], ah[X3 ? 4 : 1];
  auto load_frags = [&](int slot, auto par_c) {
    constexpr int PAR = decltype(par_c)::value;  // 0: plain operands; 1 / 2: even / odd K-tile of split operands
    const char* sb = smem + slot * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < NTW; ++i) wf[i] = *reinterpret_cast<const uint4*>(sb + w_rd + i * 16 * ROWB2);
#pragma unroll
    for (int i = 0; i < 4; ++i) (PAR == 1 ? ah[i] : af[i]) = *reinterpret_cast<const uint4*>(sb + a_rd + i * 16 * ROWB2);
  };
  auto mma_all = [&](auto par_c) {
    constexpr int PAR = decltype(par_c)::value;
    if constexpr (PAR != 1) {
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) mma_kstep<ME>(acc[nt][mt], wf[nt], af[mt]);
    }
    if constexpr (PAR != 0) {
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) mma_kstep<ME>(acc[nt][mt], wf[nt], ah[mt]);
    }
  };

  // Segment s = 2k / 2k+1 is (L_k, C_k) for group 0 (waves 0-3) and 2k+1 / 2k+2 for group 1 (waves 4-7).
  //   RAW: tile k+1 must be in LDS before segment 2k+2: every wave retires its pieces of tile k+1 with a counted
  //        vmcnt ahead of the barrier that ends odd segment 2k+1 (tiles k+2, k+3 may stay in flight: 2*PER_STAGE).
  //   WAR: tile k+3 overwrites the slot of tile k-1, last read in segment 2k-1; it is issued in segments >= 2k.
  const RowSsqReq rs_req = row_scale_request<EPI>(p, m0 + wm * 64, lane);
  if constexpr (TAPS) setup_taps(); else setup_term(0);
  const bool late = __builtin_amdgcn_readfirstlane(wave) >= 4;
  // (HALO) can this tile share the staged rows between the taps?  No sequence start inside it, and a halo of at most 128 rows
  bool shared_rows = false;
  if constexpr (HALO) {
    const int t_first = m0 % p.T;
    shared_rows = halo_rows <= 128 && (t_first + BMB <= p.T || m0 + (p.T - t_first) >= p.M);
  }
  if constexpr (HALO) {
    if (shared_rows) {  // chunk 0 (rows, halo), the weight tiles of K-tiles 0..2
      const uint32_t abase = lds_base + STAGES * TILE;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        glds16(a_ptr[i], abase + halo_rows * ROWB2 + wave * 2048 + i * 1024);
        a_ptr[i] += tap_ainc;
      }
      glds16_lanes(halo_ptr, abase + wave * (halo_rows >> 3) * ROWB2, halo_mask);
      halo_ptr += halo_inc;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(w_ptr[i] + tap_woff, lds_base + t * TILE + wave * 2048 + i * 1024);
        tap_woff += tap_wstride;
      }
      tap_woff = 0;
#pragma unroll
      for (int i = 0; i < 2; ++i) w_ptr[i] += w_inc;
      pipe_sync<4>();  // chunk 0 and the first weight tile have landed
    }
  }
  if (!shared_rows) {
#pragma unroll
  for (int st = 0; st < STAGES - 1; ++st)
    if (st < nkt) stage(st);
  if (BNB != 256 && __builtin_amdgcn_readfirstlane(wave) >= 4) {  // (tile 0 landed; these waves carry PER_LATE pieces per stage)
    if (nkt > 2) pipe_sync<2 * PER_LATE>(); else if (nkt > 1) pipe_sync<PER_LATE>(); else pipe_sync<0>();
  } else {
    if (nkt > 2) pipe_sync<2 * PER_STAGE>(); else if (nkt > 1) pipe_sync<PER_STAGE>(); else pipe_sync<0>();
  }
  }
  const float row_scale = row_scale_finish(p, rs_req);
  __builtin_amdgcn_sched_barrier(0);
  if (late) pipe_sync<63>();  // the stagger
  // The K loop is specialised on the wave's group, on whether the tile still stages (kt + 3 < nkt) and on the vmcnt its
  // counted barrier waits for: no per-iteration conditions (a uniform branch in the wave that is feeding the MFMA pipe
  // idles the pipe).
  auto ktile = [&](auto late_c, auto stage_c, auto sync_c, int slot, int fill, auto par_c) {
    constexpr bool LATE = decltype(late_c)::value, STAGE = decltype(stage_c)::value;
    constexpr int SYNC = decltype(sync_c)::value;
    // ---- L segment
    if constexpr (!(DN_GEMM_ABL & 4)) load_frags(slot, par_c);
    if constexpr (STAGE && !(DN_GEMM_ABL & 1)) stage(fill);
    if constexpr (LATE) pipe_sync<SYNC>(); else pipe_sync<63>();
    __builtin_amdgcn_sched_barrier(0);
    // ---- C segment
    if constexpr (!(DN_GEMM_ABL & 8)) __builtin_amdgcn_s_setprio(1);
    if constexpr (!(DN_GEMM_ABL & 2)) mma_all(par_c);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LATE) pipe_sync<63>(); else pipe_sync<SYNC>();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto run = [&](auto late_c) {
    using std::integral_constant;
    int slot = 0, fill = STAGES - 1, kt = 0;
    auto adv = [&]() { slot = slot == STAGES - 1 ? 0 : slot + 1; fill = fill == STAGES - 1 ? 0 : fill + 1; };
    using P0 = integral_constant<int, 0>;
    constexpr int PER = decltype(late_c)::value ? PER_LATE : PER_STAGE;  // DMA pieces this wave issues per stage
    if constexpr (X3) {
      // K-tiles come in (even, odd) pairs, nkt is even; the staging tiles 0 .. nkt-4 end on an even one, so the drain is
      // [even: stages][odd: two tiles follow][even][odd]
      using EV = integral_constant<int, 1>;
      using OD = integral_constant<int, 2>;
      if (nkt >= 4) {
        for (; kt + 4 < nkt; kt += 2) {
          ktile(late_c, std::true_type{}, integral_constant<int, 2 * PER>{}, slot, fill, EV{}); adv();
          ktile(late_c, std::true_type{}, integral_constant<int, 2 * PER>{}, slot, fill, OD{}); adv();
        }
        ktile(late_c, std::true_type{}, integral_constant<int, 2 * PER>{}, slot, fill, EV{}); adv();
        ktile(late_c, std::false_type{}, integral_constant<int, PER>{}, slot, fill, OD{}); adv();
      }
      ktile(late_c, std::false_type{}, integral_constant<int, 0>{}, slot, fill, EV{}); adv();
      ktile(late_c, std::false_type{}, integral_constant<int, 0>{}, slot, fill, OD{});
    } else {
    for (; kt + 3 < nkt; ++kt) { ktile(late_c, std::true_type{}, integral_constant<int, 2 * PER>{}, slot, fill, P0{}); adv(); }
    if (nkt >= 3) { ktile(late_c, std::false_type{}, integral_constant<int, PER>{}, slot, fill, P0{}); adv(); }  // two tiles follow
    if (nkt >= 2) { ktile(late_c, std::false_type{}, integral_constant<int, 0>{}, slot, fill, P0{}); adv(); }
    ktile(late_c, std::false_type{}, integral_constant<int, 0>{}, slot, fill, P0{});
    }
  };
  // ---- HALO form.  LDS: a 4-slot ring of weight K-tiles (16 KiB each) and two activation slots of up to 24 KiB ([H halo rows ; 256
  // tile rows] x 64 bytes); K-tile kt = tap kt % 3 of chunk kt / 3 stages the weight tile of K-tile kt + 3 (2 pieces per wave) and,
  // tap 0: the wave's two row pieces of chunk c + 1, tap 1: its share of that chunk's halo (into the slot of chunk c - 1, last
  // read by K-tile 3c - 1).  Chunk c + 1 is first read by K-tile 3c + 3, so the counted wait that closes tap 2 leaves only that
  // K-tile's own two pieces in flight; the other two waits leave everything issued after the next K-tile's weights.
  constexpr int A_RING = STAGES * TILE, A_SLOT = (256 + 128) * ROWB2;
  [[maybe_unused]] auto ktile_h = [&](auto late_c, auto tap_c, auto stage_c, auto sync_c, int slot, int fill, int aslot) {
    constexpr bool LATE = decltype(late_c)::value, STAGE = decltype(stage_c)::value;
    constexpr int TAP = decltype(tap_c)::value, SYNC = decltype(sync_c)::value;
    // ---- L segment
    {
      const char* sw = smem + slot * TILE;
      const char* sa = smem + A_RING + aslot * A_SLOT + a_rdh[TAP];
#pragma unroll
      for (int i = 0; i < NTW; ++i) wf[i] = *reinterpret_cast<const uint4*>(sw + w_rd + i * 16 * ROWB2);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const uint4*>(sa + i * 16 * ROWB2);
    }
    if constexpr (STAGE) {
      const uint32_t wbase = lds_base + fill * TILE + wave * 2048;
#pragma unroll
      for (int i = 0; i < 2; ++i) glds16(w_ptr[i] + tap_woff, wbase + i * 1024);
      if constexpr (TAP == 2) {
        tap_woff = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) w_ptr[i] += w_inc;
      } else {
        tap_woff += tap_wstride;
      }
      const uint32_t abase = lds_base + A_RING + (aslot ^ 1) * A_SLOT;
      if constexpr (TAP == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          glds16(a_ptr[i], abase + halo_rows * ROWB2 + wave * 2048 + i * 1024);
          a_ptr[i] += tap_ainc;
        }
      }
      if constexpr (TAP == 1) {
        glds16_lanes(halo_ptr, abase + wave * (halo_rows >> 3) * ROWB2, halo_mask);
        halo_ptr += halo_inc;
      }
    }
    if constexpr (LATE) pipe_sync<SYNC>(); else pipe_sync<63>();
    __builtin_amdgcn_sched_barrier(0);
    // ---- C segment
    __builtin_amdgcn_s_setprio(1);
    mma_all(std::integral_constant<int, 0>{});
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LATE) pipe_sync<63>(); else pipe_sync<SYNC>();
    __builtin_amdgcn_sched_barrier(0);
  };
  [[maybe_unused]] auto run_h = [&](auto late_c) {
    using std::integral_constant;
    using T0 = integral_constant<int, 0>; using T1 = integral_constant<int, 1>; using T2 = integral_constant<int, 2>;
    int slot = 0, fill = STAGES - 1, aslot = 0;
    auto adv = [&]() { slot = slot == STAGES - 1 ? 0 : slot + 1; fill = fill == STAGES - 1 ? 0 : fill + 1; };
    const int chunks = ktiles_per_term;  // >= 2 (launch_big)
    for (int c = 0; c + 1 < chunks; ++c) {
      ktile_h(late_c, T0{}, std::true_type{}, integral_constant<int, 6>{}, slot, fill, aslot); adv();
      ktile_h(late_c, T1{}, std::true_type{}, integral_constant<int, 7>{}, slot, fill, aslot); adv();
      ktile_h(late_c, T2{}, std::true_type{}, integral_constant<int, 2>{}, slot, fill, aslot); adv();
      aslot ^= 1;
    }
    ktile_h(late_c, T0{}, std::false_type{}, integral_constant<int, 3>{}, slot, fill, aslot); adv();
    ktile_h(late_c, T1{}, std::false_type{}, integral_constant<int, 0>{}, slot, fill, aslot); adv();
    ktile_h(late_c, T2{}, std::false_type{}, integral_constant<int, 0>{}, slot, fill, aslot);
  };
  if (!shared_rows) { if (late) run(std::true_type{}); else run(std::false_type{}); }
  if constexpr (HALO) {
    if (shared_rows) { if (late) run_h(std::true_type{}); else run_h(std::false_type{}); }
  }
  if (!late) pipe_sync<63>();

  // ---- epilogue: two 64 x 64 halves of the wave's 64 (m) x 128 (n) tile through its LDS slab
  float* ep = reinterpret_cast<float*>(smem) + wave * (64 * EP_LD);
  // the two halves are spelled out with compile-time accumulator indices: a rolled loop would index `acc` at run time
  // and push the whole accumulator file to scratch
  float sc4[4];
  slab_row_scales(row_scale, lane, sc4);
  auto half = [&](auto hc) {
    constexpr int H = decltype(hc)::value;
    constexpr int NTS = NTW - 4 * H < 4 ? NTW - 4 * H : 4;  // accumulator tiles of this half (192 columns: 4, then 2)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTS; ++nt)
        *reinterpret_cast<f32x4*>(ep + (mt * 16 + frow) * EP_LD + nt * 16 + fq * 4) = scaled(acc[H * 4 + nt][mt], sc4[mt]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wave_epilogue<EPI, std::is_same<E, BF16X3>::value, std::is_same<E, F16>::value>(p, ep, m0 + wm * 64, n0 + wn * (BNB / 2) + H * 64, g, lane, NTS * 16, row_scale >= 0.f ? ROW_PRESCALED : row_scale);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // slab reads done before the second half overwrites it
  };
  half(std::integral_constant<int, 0>{});
  half(std::integral_constant<int, 1>{});
}

// ------------------------------------------------------------------------------------------ 256 x 128 tile, two workgroups per CU
// For the short-K contractions of a transformer layer (K = 512: the q / kv and GEGLU projections, 16 K-tiles): a workgroup there is
// prologue (fill latency), a K loop no longer than both together, and a store-bound epilogue, and the one-workgroup-per-CU tiles
// (256 x 256, 256 x 352) have nothing to run beside either.  Two workgroups of the 128 x 128 tile do overlap each other but stage
// twice the bytes per flop and read 16 fragments per 32 MFMAs.  This tile keeps the two workgroups per CU at three quarters of that
// traffic: 256 (rows) x 128 (weight rows), FOUR waves of 128 x 64 (128 accumulator registers; two waves per SIMD come from the two
// workgroups), 64-byte K-tiles like the 256 x 256 kernel (row-major or K-blocked operands: a 16-row piece is 1 KiB) in a 3-stage
// ring of 24 KiB (72 KiB per workgroup, 144 per CU), one barrier per K-tile: 12 fragment reads per 32 MFMAs and 24 KiB of DMA per
// 128 MFMAs.  No stagger is scheduled: the partner workgroup is out of phase by construction after the first round of tiles.
constexpr int BNM = 128;
template <typename E, int EPI>
__global__ __launch_bounds__(256, 2) void conv_gemm_mid2_kernel(const DnGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(!std::is_same<E, BF16X3>::value, "split operands pair K-tiles: not on this tile");
  if constexpr (std::is_same<E, F16>::value) f16_saturate();
  constexpr int ES = Elem<E>::bytes;
  constexpr int KT = ROWB2 / ES;
  constexpr int BMM = 256, STAGES = 3;
  constexpr int WT = BNM * ROWB2, STAGE_BYTES = WT + BMM * ROWB2;  // 8 + 16 KiB
  constexpr int PER = 6;                                            // DMA pieces per wave per stage: 2 weight + 4 row pieces of 16 rows

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int g = blockIdx.y;
  const int np_total = p.N * (EPI == DN_EPI_GEGLU ? 2 : 1);
  const int n_tiles_n = (np_total + BNM - 1) / BNM;
  int logical;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int mt_, nt_;
  tile_coords(logical, n_tiles_n, (p.M + BMM - 1) / BMM, (p.pad_ >> 24) & 0xff, mt_, nt_);
  const int m0 = mt_ * BMM, n0 = nt_ * BNM;
  const int w_rows = (np_total + 127) / 128 * 128;

  // ---- staging: wave w stages weight rows [32 w, +32) and activation rows [64 w, +64), 16 rows x 64 B per piece
  const int srow = lane >> 2;
  const int schunk = (lane & 3) ^ (((lane >> 5) & 1) << 1);
  int a_row[4], a_t[4], w_row[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + wave * 64 + i * 16 + srow;
    m = m < p.M ? m : p.M - 1;
    a_row[i] = m;
    a_t[i] = m % p.T;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = n0 + wave * 32 + i * 16 + srow;
    w_row[i] = n < w_rows ? n : w_rows - 1;
  }
  const int ktiles_per_term = p.K / KT;
  const int nkt = p.n_terms * ktiles_per_term;
  const char* zero_src = reinterpret_cast<const char*>(g_zero_page) + schunk * 16;
  const char* a_ptr[4];
  const char* w_ptr[2];
  int a_inc[4], w_inc = ROWB2;
  int s_term = 0, s_kk = 0;
  auto setup_term = [&](int term) {
    const DnGemmTerm& tm = p.terms[term];
    const int shift = tm.shift_by_group ? tm.shift * (1 << g) : tm.shift;
    const char* A = reinterpret_cast<const char*>(tm.A) + (tm.a_gstride * g) * ES + schunk * 16;
    const char* W = reinterpret_cast<const char*>(tm.W) + (tm.w_gstride * g) * ES + schunk * 16;
    const bool a_kb = tm.layout & DN_LAYOUT_A_KBLOCKED, w_kb = tm.layout & DN_LAYOUT_W_KBLOCKED;
    const int64_t a_rowb = a_kb ? ROWB2 : (int64_t)tm.lda * ES, w_rowb = w_kb ? ROWB2 : (int64_t)(tm.ldw ? tm.ldw : p.K) * ES;
    const int a_step = a_kb ? p.M * ROWB2 : ROWB2;
    w_inc = w_kb ? w_rows * ROWB2 : ROWB2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool valid = shift_valid(a_t[i], shift, p.T);
      a_ptr[i] = valid ? A + (int64_t)(a_row[i] - shift) * a_rowb : zero_src;
      a_inc[i] = valid ? a_step : 0;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) w_ptr[i] = W + (int64_t)w_row[i] * w_rowb;
  };
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lptr_t)smem);
  auto stage = [&](int slot) {
    const uint32_t wbase = lds_base + slot * STAGE_BYTES + wave * 2048;
    const uint32_t abase = lds_base + slot * STAGE_BYTES + WT + wave * 4096;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(w_ptr[i], wbase + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(a_ptr[i], abase + i * 1024);
    if (++s_kk == ktiles_per_term) {
      s_kk = 0;
      if (++s_term < p.n_terms) setup_term(s_term);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) a_ptr[i] += a_inc[i];
#pragma unroll
      for (int i = 0; i < 2; ++i) w_ptr[i] += w_inc;
    }
  };

  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  const int coff = (fq ^ (((frow >> 3) & 1) << 1)) << 4;
  const int w_rd = (wn * 64 + frow) * ROWB2 + coff;
  const int a_rd = WT + (wm * 128 + frow) * ROWB2 + coff;
  uint4 wf[4], af[8];

  setup_term(0);
  stage(0);
  if (nkt > 1) stage(1);
  int slot = 0, fill = 2;
  for (int kt = 0; kt < nkt; ++kt) {
    // tile kt landed for every wave (tile kt + 1 may stay in flight); every wave is done reading tile kt - 1, whose slot receives tile kt + 2
    if (kt + 1 < nkt) pipe_sync<PER>(); else pipe_sync<0>();
    if (kt + 2 < nkt) stage(fill);
    const char* sb = smem + slot * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const uint4*>(sb + w_rd + i * 16 * ROWB2);
#pragma unroll
    for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const uint4*>(sb + a_rd + i * 16 * ROWB2);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) mma_kstep<E>(acc[nt][mt], wf[nt], af[mt]);
    slot = slot == STAGES - 1 ? 0 : slot + 1;
    fill = fill == STAGES - 1 ? 0 : fill + 1;
  }
  __syncthreads();  // ring reads over: the slabs overwrite it

  // ---- epilogue: the wave's 128 (m) x 64 (n) sub-tile as two 64 x 64 slabs through its private LDS slab
  float* ep = reinterpret_cast<float*>(smem) + wave * (64 * EP_LD);
  auto half = [&](auto hc) {
    constexpr int H = decltype(hc)::value;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        *reinterpret_cast<f32x4*>(ep + (mt * 16 + frow) * EP_LD + nt * 16 + fq * 4) = acc[nt][H * 4 + mt];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wave_epilogue<EPI, false, std::is_same<E, F16>::value>(p, ep, m0 + wm * 128 + H * 64, n0 + wn * 64, g, lane, 64, -1.f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  half(std::integral_constant<int, 0>{});
  half(std::integral_constant<int, 1>{});
}

// ------------------------------------------------------------------------------------------ whole-row tile + fused RMSNorm
// For the two contractions per transformer layer that close a residual branch (to_out and the FFN's last Linear, N = D)
// and the WaveNet's final 1x1 conv: a 64 (rows) x 512 (all columns) tile, so one workgroup sees complete rows of the new
// residual stream and can emit the next block's RMSNorm (adaptive gamma/beta or learned gamma) in the same launch.  That
// removes the separate norm kernel -- one fp32 read + one bf16 write of the stream per norm, 25 per denoising step.
// 8 waves, wave w owns columns [64w, 64w+64) of all 64 rows; 2-stage ring of 72 KiB (weights 64 KiB + rows 8 KiB).
template <typename E, int EPI>
__global__ __launch_bounds__(512, 1) void conv_gemm_row_kernel(const DnGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if constexpr (std::is_same<E, F16>::value) f16_saturate();  // stores of a value beyond 65504 clamp instead of becoming inf
  constexpr int ES = Elem<E>::bytes;
  constexpr int KT = ROWB / ES;
  constexpr int A_TILE = 64 * ROWB, W_TILE = 512 * ROWB, STAGE_BYTES = W_TILE + A_TILE;
  constexpr int PART_OFF = 8 * 64 * EP_LD * 4;  // per-row partial sums of squares live behind the 8 slabs

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.y;
  const int m0 = blockIdx.x * 64;
  const int w_rows = (p.N + 127) / 128 * 128;

  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ srow;
  int a_row, a_t, w_row[8];
  {
    int m = m0 + wave * 8 + srow;
    m = m < p.M ? m : p.M - 1;
    a_row = m;
    a_t = m % p.T;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int n = wave * 64 + i * 8 + srow;
      w_row[i] = n < w_rows ? n : w_rows - 1;
    }
  }
  const int ktiles_per_term = p.K / KT;
  const int nkt = p.n_terms * ktiles_per_term;
  const char* zero_src = reinterpret_cast<const char*>(g_zero_page) + schunk * 16;
  const char* a_ptr;
  const char* w_ptr[8];
  int a_inc;
  int s_term = 0, s_kk = 0;
  auto setup_term = [&](int term) {
    const DnGemmTerm& tm = p.terms[term];
    const int shift = tm.shift_by_group ? tm.shift * (1 << g) : tm.shift;
    const char* A = reinterpret_cast<const char*>(tm.A) + (tm.a_gstride * g) * ES + schunk * 16;
    const char* W = reinterpret_cast<const char*>(tm.W) + (tm.w_gstride * g) * ES + schunk * 16;
    const bool valid = shift_valid(a_t, shift, p.T);
    a_ptr = valid ? A + (int64_t)(a_row - shift) * tm.lda * ES : zero_src;
    a_inc = valid ? ROWB : 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) w_ptr[i] = W + (int64_t)w_row[i] * p.K * ES;
  };
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lptr_t)smem);
  auto stage = [&](int slot) {
    const uint32_t sbase = lds_base + slot * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 8; ++i) glds16(w_ptr[i], sbase + (wave * 8 + i) * 1024);
    glds16(a_ptr, sbase + W_TILE + wave * 1024);
    if (++s_kk == ktiles_per_term) {
      s_kk = 0;
      if (++s_term < p.n_terms) setup_term(s_term);
    } else {
      a_ptr += a_inc;
#pragma unroll
      for (int i = 0; i < 8; ++i) w_ptr[i] += ROWB;
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fq = lane >> 4;
  const int sw = frow & 7;
  const int w_rd = (wave * 64 + frow) * ROWB;
  const int a_rd = W_TILE + frow * ROWB;

  setup_term(0);
  stage(0);
  int slot = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    pipe_sync<0>();
    if (kt + 1 < nkt) stage(slot ^ 1);
    const char* sb = smem + slot * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + fq) ^ sw) << 4;
      uint4 wf[4], af[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const uint4*>(sb + w_rd + i * 16 * ROWB + coff);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const uint4*>(sb + a_rd + i * 16 * ROWB + coff);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) mma_kstep<E>(acc[nt][mt], wf[nt], af[mt]);
    }
    slot ^= 1;
  }
  __syncthreads();  // ring reads over: the slabs may overwrite it

  // ---- epilogue 1: transpose, residual / positional add, fp32 stream store, per-row partial sums of squares
  float* ep = reinterpret_cast<float*>(smem) + wave * (64 * EP_LD);
  float* part = reinterpret_cast<float*>(smem + PART_OFF);  // [64 rows][8 waves]
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      *reinterpret_cast<f32x4*>(ep + (mt * 16 + frow) * EP_LD + nt * 16 + fq * 4) = acc[nt][mt];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  const int c4 = (lane & 15) * 4;
  const int n = wave * 64 + c4;
  const bool col_ok = n < p.N;
  const float* bias = p.bias ? p.bias + p.bias_gstride * g : nullptr;
  const float4 bv = (bias && col_ok) ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0, 0, 0, 0);
  float* out = reinterpret_cast<float*>(p.out) + p.out_gstride * g;  // RESADD / POSEMB always write the fp32 stream
  const float* resb = EPI == DN_EPI_RESADD ? reinterpret_cast<const float*>(p.res) + p.res_gstride * g : nullptr;
  int b0 = 0, t0 = 0;
  {
    const int m_first = m0 + (lane >> 4);
    b0 = m_first / p.T;
    t0 = m_first - b0 * p.T;
  }
  int bs[16];  // sequence of each of this lane's rows (for the adaptive norm rows)
  float4 xv[16];
#pragma unroll
  for (int jb = 0; jb < 16; jb += 8) {
    float4 rv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + (jb + i) * 4 + (lane >> 4);
      rv[i] = make_float4(0, 0, 0, 0);
      bs[jb + i] = b0;
      if (m < p.M && col_ok) {
        if constexpr (EPI == DN_EPI_RESADD) {
          rv[i] = *reinterpret_cast<const float4*>(resb + (int64_t)m * p.ldr + n);
        } else {
          const int pos = t0 < p.lengths[b0] ? t0 + 1 : 0;
          rv[i] = *reinterpret_cast<const float4*>(p.pos_table + (int64_t)pos * p.pos_ld + n);
        }
      }
      if (p.T >= 4) {
        t0 += 4;
        if (t0 >= p.T) {
          t0 -= p.T;
          ++b0;
        }
      } else {
        b0 = (m + 4) / p.T;
        t0 = (m + 4) - b0 * p.T;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = (jb + i) * 4 + (lane >> 4);
      const int m = m0 + row;
      const float4 a4 = *reinterpret_cast<const float4*>(ep + row * EP_LD + c4);
      float4 v = make_float4(a4.x + bv.x + rv[i].x, a4.y + bv.y + rv[i].y, a4.z + bv.z + rv[i].z, a4.w + bv.w + rv[i].w);
      if (!col_ok) v = make_float4(0, 0, 0, 0);
      xv[jb + i] = v;
      if (m < p.M && col_ok) *reinterpret_cast<float4*>(out + (int64_t)m * p.ldo + n) = v;
      float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;  // columns >= norm_D are exact zeros (zero-padded weights)
      ss += __shfl_xor(ss, 1, 16);
      ss += __shfl_xor(ss, 2, 16);
      ss += __shfl_xor(ss, 4, 16);
      ss += __shfl_xor(ss, 8, 16);
      if ((lane & 15) == 0) part[row * 8 + wave] = ss;
    }
  }
  if (!p.norm_out) return;
  __syncthreads();

  // ---- epilogue 2: the next block's RMSNorm of the rows just produced
  const float scale = sqrtf((float)p.norm_D);
  const bool in_d = n < p.norm_D;
  const float4 gam = (p.norm_gamma && in_d) ? *reinterpret_cast<const float4*>(p.norm_gamma + n) : make_float4(1, 1, 1, 1);
  const bool nbf = dn_is16(p.norm_dtype);  // (the kernel's own 2-byte type)
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int row = j * 4 + (lane >> 4);
    const int m = m0 + row;
    if (m >= p.M || n >= p.norm_ld) continue;
    const float4 p0 = *reinterpret_cast<const float4*>(part + row * 8);
    const float4 p1 = *reinterpret_cast<const float4*>(part + row * 8 + 4);
    const float denom = fmaxf(sqrtf(p0.x + p0.y + p0.z + p0.w + p1.x + p1.y + p1.z + p1.w), 1e-12f);
    float4 y = make_float4(xv[j].x / denom * scale * gam.x, xv[j].y / denom * scale * gam.y, xv[j].z / denom * scale * gam.z,
                           xv[j].w / denom * scale * gam.w);
    if (p.norm_gb && in_d) {
      const float* gr = p.norm_gb + (int64_t)bs[j] * p.norm_gb_ld + n;
      const float4 ga = *reinterpret_cast<const float4*>(gr);
      const float4 be = *reinterpret_cast<const float4*>(gr + p.norm_gb_half);
      y = make_float4(y.x * ga.x + be.x, y.y * ga.y + be.y, y.z * ga.z + be.z, y.w * ga.w + be.w);
    }
    if (!in_d) y = make_float4(0, 0, 0, 0);
    if (nbf) store4t<std::is_same<E, F16>::value ? 3 : 1>(p.norm_out, (int64_t)m * p.norm_ld + n, y.x, y.y, y.z, y.w);
    else store4t<0>(p.norm_out, (int64_t)m * p.norm_ld + n, y.x, y.y, y.z, y.w);
  }
}

// ------------------------------------------------------------------------------------------ 256 x 352 "fat" tile
// The FFN's causal conv (N = K/3 = 1408 padded inner width) is 45 % of a denoising step's FLOPs, and 1408 = 4 x 352:
// a 256 (rows) x 352 (columns) tile covers [16384 x 1408] with exactly 64 x 4 = 256 workgroups -- one round on the 256
// CUs with no ragged last column tile (the 256 x 256 tiling needs 384 workgroups = 1.5 rounds and pads N to 1536).
// Four waves, ONE PER SIMD, each a 128 (m) x 176 (n) sub-tile = 88 accumulator tiles (352 registers of the 512 a lone
// wave may use): per 32-deep K-tile a wave issues 88 MFMAs against 19 fragment reads and ~10 DMA pieces, so the LDS and
// the L2->LDS path run at a fraction of their rates and the MFMA pipe is the only busy resource.  With no partner wave
// the overlap is inside the instruction stream: weight fragments stream one n-tile ahead of the MFMAs that use them,
// the next K-tile's activation fragments are fetched under the last two n-tiles, the DMA runs three K-tiles ahead in
// a 4-stage 152 KiB ring, and there is one barrier per K-tile, placed mid-stream.
// 88 accumulator tiles are 352 registers: more than the 256 AGPRs.  The compiler keeps every MFMA accumulator of a
// function in one register class and would shuttle the overflow through spare AGPRs (accvgpr moves + MFMA-result
// nops on the critical path), so the tiles are pinned by hand: n-tiles 0..7 (64 tiles) in AGPRs, n-tiles 8..10
// (24 tiles) in ordinary VGPRs, each MFMA written with the matching constraint.  Hazards the compiler can no longer
// see: an accumulator is touched once per K-tile (88 MFMAs apart), operands come from LDS behind s_waitcnt, and the
// epilogue's first read is separated from the last MFMA by explicit s_nops.
//
// The K loop of that kernel is scheduled by hand as well.  A lone wave hides LDS latency only inside its own
// instruction stream; the compiler's s_waitcnt placement drains the LDS queue at the loop header (it cannot prove what
// is pending across the back edge), which exposes a full loaded-LDS latency per K-tile.  So the fragment reads are
// inline asm too (the compiler then tracks nothing), every use is preceded by an explicit counted s_waitcnt that
// carries the registers it guards as in/out operands (a true dependence: no consumer can be moved above it), and all
// of these statements are `volatile`, which keeps them in program order.  LDS returns data in order, so "at most N
// younger requests outstanding" is exact; a pending scalar load only makes a wait longer, never shorter.
#ifndef DN_FAT_ABL
#define DN_FAT_ABL 0
#endif
typedef __attribute__((ext_vector_type(2))) unsigned long u32x4;  // a 128-bit fragment as two 64-bit halves (4 VGPRs)

template <bool IN_AGPR, bool H16>  // H16: IEEE-half operands (same shape, same rate)
__device__ __forceinline__ void mma_pinned_bf16(f32x4& acc, const u32x4& w, const u32x4& a) {
  if constexpr (H16) {
    if constexpr (IN_AGPR) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(a));
  } else {
    if constexpr (IN_AGPR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(a));
  }
}

template <int OFFSET>
__device__ __forceinline__ void lds_request(u32x4& dst, uint32_t addr) {
  static_assert(OFFSET >= 0 && OFFSET < 65536, "ds_read offset field");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFFSET));
}

template <int N>
__device__ __forceinline__ void lds_wait(u32x4& r) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lds_wait_all_but() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N)); }

// LDS-request bookkeeping of the one-wave-per-SIMD K-tile (NT n-tiles of 8 MFMAs).  Program order of the requests of
// n-tile g: the weight fragment that will be used two n-tiles later, then -- for n-tiles BG .. BG + 8/NPG - 1 -- NPG of the
// next K-tile's 8 activation fragments.  LDS returns in order, so a use must allow exactly the requests issued after
// the one it needs to stay outstanding; these functions count them (two consecutive K-tiles laid end to end).
// An accumulator tile from its AGPRs to LDS (DS instructions take AGPR data operands on gfx90a and later).
template <int OFF>
__device__ __forceinline__ void acc_to_lds(uint32_t lds_addr, const f32x4& v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(lds_addr), "a"(v), "n"(OFF) : "memory");
}

template <int NT, int BG, int NPG, int MT = 8>
struct SoloSched {
  static constexpr int per_group(int g) { return 1 + ((g >= BG && g < BG + MT / NPG) ? NPG : 0); }
  static constexpr int prefix(int g) { int n = 0; for (int h = 0; h < g; ++h) n += per_group(h); return n; }
  static constexpr int R = prefix(NT);
  // weight fragment nt of a K-tile is requested in n-tile nt-2 of the same K-tile, or (nt = 0, 1) in n-tile NT-2+nt of
  // the previous one; it is waited for at the top of n-tile nt
  static constexpr int younger_w(int nt) { return (R + prefix(nt)) - ((nt >= 2 ? R + prefix(nt - 2) : prefix(NT - 2 + nt)) + 1); }
  // the last activation fragment is the last request of n-tile BG + 8/NPG - 1; the copies that need it sit in the last
  // n-tile behind that n-tile's own weight request
  static constexpr int younger_copy() { return (prefix(NT - 1) + 1) - ((prefix(BG + MT / NPG - 1) + per_group(BG + MT / NPG - 1) - 1) + 1); }
};
static_assert(SoloSched<11, 1, 1>::younger_w(0) == 1 && SoloSched<11, 1, 1>::younger_w(2) == 2 && SoloSched<11, 1, 1>::younger_w(5) == 3 &&
              SoloSched<11, 1, 1>::younger_w(10) == 2 && SoloSched<11, 1, 1>::younger_copy() == 2, "schedule arithmetic");
// register copy that stays behind the wait above it (a plain C++ copy could be scheduled ahead of the wait and read a
// register whose LDS data is still in flight)
__device__ __forceinline__ void copy_after_wait(u32x4& dst, const u32x4& src) {
  asm volatile("v_mov_b64 %0, %1" : "=v"(dst.x) : "v"(src.x));
  asm volatile("v_mov_b64 %0, %1" : "=v"(dst.y) : "v"(src.y));
}
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// NTW = n-tiles per wave: 11 -> the 256 x 352 tile (N a multiple of 352, BIAS epilogue), 8 -> a 256 x 256 tile for any N and
// every epilogue (64 accumulator tiles, all in AGPRs).
// WAVES = 8 (NTW = 8 only) runs the same instruction stream with TWO waves per SIMD, each on a 64 x 128 sub-tile: the partner's
// MFMAs cover a wave's DMA-issue stalls without the segment barriers of the staggered two-group kernels.
// HALO (256 x 352 tile, three taps of one causal conv with shifts 2d, d, 0, d <= 8; tiles without a sequence start inside them, the
// others run the shifted-copies loop): the activation rows of a 32-element K-chunk are staged ONCE, as the tile's 256 rows
// plus the 16 rows in front of them (zeros for frames in front of the sequence), and the three taps read them at row offsets
// 16 - shift -- instead of three shifted copies of the same rows.  A chunk then moves 17 + 3 x 22 KiB into LDS instead of
// 3 x (16 + 22): the L2 -> LDS path, which this loop keeps ~70 % busy, carries 27 % less and each wave issues 23 instead of 30 DMA
// pieces per chunk.  Same K order as TAPS_INNER (chunk-major, tap-minor), so the same bits.
template <typename E, int EPI, bool TAPS_INNER, int NTW, int WAVES, bool HALO = false>
__global__ __launch_bounds__(64 * WAVES, 1) void conv_gemm_fat_kernel(const DnGemmParams p) {
  static_assert(IsHalf<E>::value, "the hand-scheduled tiles are built for 2-byte operands only");
  constexpr bool H16 = std::is_same<E, F16>::value;
  static_assert(!HALO || (TAPS_INNER && NTW == 11 && WAVES == 4), "the halo staging is built for the 256 x 352 tile in tap-inner order");
  static_assert(NTW == 11 || NTW == 8, "n-tiles per wave");
  static_assert(WAVES == 4 || (WAVES == 8 && NTW == 8), "4 waves (one per SIMD) or, on the 256 x 256 tile, 8");
  static_assert(NTW == 8 || EPI == DN_EPI_BIAS || EPI == DN_EPI_GEGLU,
                "the 352-wide tile carries the BIAS and GEGLU epilogues (its waves start at multiples of 176 columns: fine for "
                "GEGLU's self-contained 16-column tiles, not for the others' assumptions)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if constexpr (std::is_same<E, F16>::value) f16_saturate();  // stores of a value beyond 65504 clamp instead of becoming inf
  constexpr int ES = Elem<E>::bytes;
  constexpr int KT = ROWB2 / ES;
  constexpr int BMF = 256, BNF = 32 * NTW, STAGES = 4;
  constexpr int W_BYTES = BNF * ROWB2, A_BYTES = BMF * ROWB2, STAGE_BYTES = W_BYTES + A_BYTES;  // 352: 22528 + 16384
  constexpr int NWP = BNF / 16, WPW = (NWP + WAVES - 1) / WAVES;  // weight pieces per stage / per wave (352: 22 / 6, two waves repeat one)
  constexpr int APW = 16 / WAVES;   // row pieces (16 rows each) per wave
  constexpr int PER = WPW + APW;    // DMA pieces per wave per stage: WPW weight pieces, then APW of the 16 row pieces
  constexpr int NT = NTW, MT = 32 / WAVES;  // a wave's sub-tile: 16 MT rows (128 or 64) x 16 NT columns
  constexpr int RPW = 16 * MT;
  constexpr int BG = NTW == 11 ? 1 : 0;   // n-tile whose top carries the barrier: the first one that touches K-tile kt+1
  constexpr int NPG = (NTW == 11 || WAVES == 8) ? 1 : 2;  // next-K-tile activation fragments requested per n-tile (n-tiles BG .. BG + MT/NPG - 1)
  using Sched = SoloSched<NT, BG, NPG, MT>;
  static_assert((NT - 2) % 3 == 0, "weight fragments 2..NT-1 must cycle the 3-deep ring a whole number of times per K-tile");
  static_assert(BG + PER - 1 <= NT - 1 && BG + MT / NPG - 1 <= NT - 3, "DMA pieces and activation requests fit the n-tiles");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int g = blockIdx.y;
  const int np_total = p.N * (EPI == DN_EPI_GEGLU ? 2 : 1);  // packed weight rows that carry output
  const int n_tiles_n = (np_total + BNF - 1) / BNF;
  int logical;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int m0 = (logical / n_tiles_n) * BMF;
  const int n0 = (logical % n_tiles_n) * BNF;
  // the packed weight has rows up to the next multiple of 128: pieces of a ragged last tile beyond that re-read its
  // last piece (their columns are never stored)
  const int pc_max = min(NWP, ((np_total + 127) / 128 * 128 - n0) / 16) - 1;

  // ---- staging: 16-row x 64-byte pieces; wave w takes row pieces 4w..4w+3 and weight pieces w, w+4, .., (w+20 or 21).
  // Weight pieces differ by a uniform row offset: one per-lane 32-bit offset + a scalar base per piece (saddr form).
  // Row pieces keep per-lane 64-bit pointers (a lane whose frame precedes the sequence start reads the zero page).
  //
  // K-tile order.  General case: term-outer (all K-tiles of term 0, then term 1, ..).  When the terms are the taps of
  // ONE causal conv (same activation tensor) the TAPS_INNER instantiation (opt-in, see launch_fat) runs tap-inner: K-tile n is tap
  // n % n_terms of K-chunk n / n_terms, so the three taps read (nearly) the same activation rows back to back and the
  // XCD's L2 serves two of the three reads; term-outer re-fetches the XCD's 5.8 MB activation panel through the
  // fabric once per tap (measured 234 MB of fabric reads per FFN-conv launch = 3 x 46 MB + 8 XCDs x 11.9 MB of weights).
  const int srow = lane >> 2;
  const int schunk = (lane & 3) ^ (((lane >> 5) & 1) << 1);
  const int ktiles_per_term = p.K / KT;
  const int nkt = p.n_terms * ktiles_per_term;
  constexpr bool taps_inner = TAPS_INNER;  // compile-time: a uniform branch in this K loop costs the lone wave ~20 cycles
  const char* a_ptr[APW];  // term-outer: this term's (shifted) row; tap-inner: the unshifted row
  int a_inc[APW];          // term-outer: 64 or 0 (zero page); tap-inner: the row's frame index m % T
  uint32_t w_voff = 0;
  uint64_t w_base = 0;   // uniform: term weight base + tile's first row (term-outer)
  uint32_t w_step = ROWB2;  // bytes from a K-tile of the weight to the next: 64 along a row, or a whole [rows][32] block (K-blocked)
  uint64_t piece_stride = (uint64_t)16 * p.K * ES;  // bytes between weight pieces (16 rows)
  const int w_rows = (np_total + 127) / 128 * 128;  // rows of the packed weight
  int s_term = 0, s_kk = 0;
  const char* const zero_src = reinterpret_cast<const char*>(g_zero_page) + schunk * 16;
  auto setup_term = [&](int term) {
    const DnGemmTerm& tm = p.terms[term];
    const int shift = tm.shift_by_group ? tm.shift * (1 << g) : tm.shift;
    const char* A = reinterpret_cast<const char*>(tm.A) + (tm.a_gstride * g) * ES + schunk * 16;
    // K-blocked operands ([K/32][rows][32], DN_LAYOUT_*): a 16-row piece is 1 KiB of whole cache lines, the next K-tile a block away
    const bool a_kb = tm.layout & DN_LAYOUT_A_KBLOCKED, w_kb = tm.layout & DN_LAYOUT_W_KBLOCKED;
    const int64_t a_row = a_kb ? ROWB2 : (int64_t)tm.lda * ES;
    const int a_step = a_kb ? p.M * ROWB2 : ROWB2;
#pragma unroll
    for (int i = 0; i < APW; ++i) {
      int m = m0 + (wave * APW + i) * 16 + srow;
      m = m < p.M ? m : p.M - 1;
      const bool valid = shift_valid(m % p.T, shift, p.T);
      a_ptr[i] = valid ? A + (int64_t)(m - shift) * a_row : zero_src;
      a_inc[i] = valid ? a_step : 0;
    }
    const int w_row = w_kb ? ROWB2 : p.K * ES;
    w_base = (uint64_t)(uintptr_t)tm.W + (uint64_t)tm.w_gstride * g * ES + (uint64_t)n0 * w_row;
    w_voff = (uint32_t)(srow * w_row + schunk * 16);
    w_step = w_kb ? (uint32_t)w_rows * ROWB2 : ROWB2;
    piece_stride = (uint64_t)16 * w_row;
  };
  // tap-inner: the per-tap uniforms are affine in the tap index (launch_fat checks it: shift_t = shift_0 - t * step,
  // W_t = W_0 + t * stride), so they are carried as running scalars -- no per-tap table (a table indexed by the
  // run-time tap index would live in scratch, whose loads share vmcnt with the DMA)
  int tap_shift = 0, tap_shift0 = 0, tap_sstep = 0;
  uint32_t tap_delta = 0, tap_delta0 = 0, tap_dstep = 0;  // bytes between the unshifted row and the tap's row
  uint64_t tap_wbase = 0, tap_wbase0 = 0;
  int64_t tap_wstride = 0;
  int tap_akinc = ROWB2;  // bytes from a K-chunk of an activation row to the next
  const char* halo_ptr = nullptr;  // HALO: this lane's row of the 16 rows in front of the tile (lanes 0..15), or the zero page
  int halo_inc = 0;
  auto setup_taps = [&]() {
    const DnGemmTerm& t0 = p.terms[0];
    const DnGemmTerm& t1 = p.terms[1];
    const int sh0 = t0.shift_by_group ? (t0.shift << g) : t0.shift, sh1 = t1.shift_by_group ? (t1.shift << g) : t1.shift;
    // K-blocked operands ([K/32][rows][32], the same layout in every tap: launch_fat checks): rows 64 bytes apart, the next K-chunk a block away
    const bool a_kb = t0.layout & DN_LAYOUT_A_KBLOCKED, w_kb = t0.layout & DN_LAYOUT_W_KBLOCKED;
    const uint32_t a_rowb = a_kb ? (uint32_t)ROWB2 : (uint32_t)(t0.lda * ES);
    const uint32_t w_rowb = w_kb ? (uint32_t)ROWB2 : (uint32_t)(p.K * ES);
    tap_shift0 = sh0; tap_sstep = sh0 - sh1;
    tap_delta0 = (uint32_t)sh0 * a_rowb; tap_dstep = (uint32_t)tap_sstep * a_rowb;
    tap_wbase0 = (uint64_t)(uintptr_t)t0.W + (uint64_t)t0.w_gstride * g * ES + (uint64_t)n0 * w_rowb;
    tap_wstride = (int64_t)((uintptr_t)t1.W - (uintptr_t)t0.W);
    tap_shift = tap_shift0; tap_delta = tap_delta0; tap_wbase = tap_wbase0;
    tap_akinc = a_kb ? p.M * ROWB2 : ROWB2;
    w_step = w_kb ? (uint32_t)w_rows * ROWB2 : ROWB2;
    piece_stride = (uint64_t)16 * w_rowb;
    const char* A = reinterpret_cast<const char*>(t0.A) + (t0.a_gstride * g) * ES + schunk * 16;
#pragma unroll
    for (int i = 0; i < APW; ++i) {
      int m = m0 + (wave * APW + i) * 16 + srow;
      m = m < p.M ? m : p.M - 1;
      a_ptr[i] = A + (int64_t)m * a_rowb;
      a_inc[i] = m % p.T;
    }
    w_voff = (uint32_t)(srow * w_rowb + schunk * 16);
    if constexpr (HALO) {  // lanes 0..15 of wave w: rows m0 - 16 + 4 w + (lane >> 2), which land in LDS rows 4 w + (lane >> 2) of the halo piece
      const int hchunk = (lane & 3) ^ ((wave >> 1) << 1);  // the swizzle of LDS rows 8..15
      const int hrow = wave * 4 + ((lane >> 2) & 3) - 16;   // row relative to the tile's first one
      const bool before = m0 % p.T + hrow < 0;              // a frame in front of the tile's sequence: zeros
      const char* A0 = reinterpret_cast<const char*>(t0.A) + (t0.a_gstride * g) * ES + hchunk * 16;
      halo_ptr = before ? reinterpret_cast<const char*>(g_zero_page) + hchunk * 16 : A0 + (int64_t)(m0 + hrow) * a_rowb;
      halo_inc = before ? 0 : tap_akinc;
    }
  };
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lptr_t)smem);
  // one of the wave's PER DMA pieces of a stage: 0..5 weight pieces, 6..9 row pieces.  Two halves (M0 <- LDS address,
  // then the load) so that the K loop can put them into different MFMA gaps; stage_piece = both, back to back.
  auto piece_lds_addr = [&](auto i_c, int slot) -> uint32_t {
    constexpr int i = decltype(i_c)::value;
    const uint32_t sbase = lds_base + slot * STAGE_BYTES;
    if constexpr (i < WPW) {
      int pc = wave + WAVES * i;
      pc = pc < pc_max ? pc : pc_max;  // 352: waves 2, 3 repeat the last piece (every wave issues PER pieces); ragged N
      return sbase + pc * 1024;
    } else {
      return sbase + W_BYTES + (wave * APW + (i - WPW)) * 1024;
    }
  };
  uint64_t piece_base = 0;  // SALU copy of the current weight piece's scalar base (see glds_set_m0_base)
  auto piece_setup = [&](auto i_c, int slot) {
    constexpr int i = decltype(i_c)::value;
    if constexpr (i < WPW) {
      int pc = wave + WAVES * i;
      pc = pc < pc_max ? pc : pc_max;
      const uint64_t wb = taps_inner ? tap_wbase : w_base;
      piece_base = glds_set_m0_base(piece_lds_addr(i_c, slot), wb + pc * piece_stride);
    } else {
      glds_set_m0(piece_lds_addr(i_c, slot));
    }
  };
  auto piece_go = [&](auto i_c) {
    constexpr int i = decltype(i_c)::value;
    if constexpr (i < WPW) {
      glds_go_s(w_voff, piece_base);
    } else {
      constexpr int j = i - WPW;
      const char* src = a_ptr[j];
      if constexpr (taps_inner) src = a_inc[j] >= tap_shift ? src - tap_delta : zero_src;  // branch-free
      glds_go(src);
    }
  };
  auto stage_piece = [&](auto i_c, int slot) {
    piece_setup(i_c, slot);
    asm volatile("s_nop 0");
    piece_go(i_c);
  };
  auto stage_advance = [&]() {
    if constexpr (taps_inner) {  // branch-free: scalar selects, and a K-chunk increment that is 0 until the taps wrap
      const bool wrap = s_term + 1 == p.n_terms;
      s_term = wrap ? 0 : s_term + 1;
      tap_shift = wrap ? tap_shift0 : tap_shift - tap_sstep;
      tap_delta = wrap ? tap_delta0 : tap_delta - tap_dstep;
      tap_wbase = wrap ? tap_wbase0 : tap_wbase + tap_wstride;
      const int inc = wrap ? tap_akinc : 0;
#pragma unroll
      for (int i = 0; i < APW; ++i) a_ptr[i] += inc;
      w_voff += wrap ? w_step : 0u;
    } else {
      if (++s_kk == ktiles_per_term) {
        s_kk = 0;
        if (++s_term < p.n_terms) setup_term(s_term);
      } else {
#pragma unroll
        for (int i = 0; i < APW; ++i) a_ptr[i] += a_inc[i];
        w_voff += w_step;
      }
    }
  };
  auto stage = [&](int slot) {
    static_for<PER>([&](auto i_c) { stage_piece(i_c, slot); });
    stage_advance();
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  const int coff = (fq ^ (((frow >> 3) & 1) << 1)) << 4;
  const uint32_t w_rd = lds_base + (wn * (16 * NTW) + frow) * ROWB2 + coff;     // + slot * STAGE_BYTES + nt * 1024
  const uint32_t a_rd = lds_base + W_BYTES + (wm * RPW + frow) * ROWB2 + coff;  // + slot * STAGE_BYTES + mt * 1024

  // One K-tile = 11 n-tiles of 8 MFMAs.  With the LDS ~2/3 busy (76 KiB of fragment reads + 38 KiB of DMA writes per
  // K-tile) its latency is several hundred cycles, so every fragment is requested long before its first use:
  //   * weight fragments run TWO n-tiles (256 MFMA cycles) ahead: fragments 0 and 1 of a K-tile live in wa / wb
  //     (requested under n-tiles 9 and 10 of the previous K-tile), fragments 2..10 rotate through the ring wr[3]
  //     (9 = 3 x 3 uses, so the ring phase is the same in every K-tile and every register index is a constant);
  //   * the next K-tile's 8 activation fragments are requested one per n-tile under n-tiles 1..8 into `nxt` and
  //     copied to `cur` in the gaps of the last n-tile, fragment mt right behind the last MFMA that reads cur[mt].
  // Requests per n-tile, in order: [weight fragment nt+2] [nxt[nt-1] if 1 <= nt <= 8]; the counted waits below follow.
  // Everything but the MFMAs is dealt out into the gaps between them (see the n-tile body).
  // The barrier sits before n-tile 1, the first point that touches tile kt+1:
  //   RAW: tile kt+1 is read only after every wave's counted vmcnt for it and that barrier.
  //   WAR: the DMA of tile kt+3 reuses the slot of tile kt-1 and is issued after the barrier of iteration kt, which
  //        every wave reaches only after it has finished iteration kt-1.
  // (After the last K-tile the cross-tile requests read a stale slot; the values are never used.)
  u32x4 wa, wb, wr[3], cur[MT], nxt[MT];
  // A uniform branch costs this lone wave ~20 cycles (its instruction stream is the only thing feeding the MFMA pipe), so
  // the K-tile body is straight-line: whether it stages (all K-tiles but the last three) and which vmcnt it waits for
  // are compile-time; the main loop runs the staging form, the last (up to) three K-tiles run the draining forms.
  auto ktile = [&](int slot, auto stage_c, auto sync_c) {
    constexpr bool STAGE = decltype(stage_c)::value;
    constexpr int SYNC = decltype(sync_c)::value;
    const int nslot = slot == STAGES - 1 ? 0 : slot + 1;
    const uint32_t w_cur = w_rd + slot * STAGE_BYTES, w_nxt = w_rd + nslot * STAGE_BYTES, a_nxt = a_rd + nslot * STAGE_BYTES;
    static_for<NT>([&](auto nt_c) {
      constexpr int nt = decltype(nt_c)::value;
      if constexpr (nt == BG) pipe_sync<SYNC>();
      u32x4& w = [&]() -> u32x4& {
        if constexpr (nt == 0) return wa;
        else if constexpr (nt == 1) return wb;
        else return wr[(nt - 2) % 3];
      }();
      lds_wait<Sched::younger_w(nt)>(w);
      // 8 MFMAs; after MFMA g the wave has ~8 issue cycles before the pipe can take the next one: one short instruction
      // per gap is free, so the n-tile's other work is dealt out one piece per gap instead of being bunched in front
      const int fill = slot == 0 ? STAGES - 1 : slot - 1;  // the slot tile kt-1 lived in receives tile kt+3
      constexpr bool dma = nt >= BG && nt - BG < PER && STAGE && !(DN_FAT_ABL & 1);
      constexpr bool dma_last = dma && nt - BG == PER - 1;  // the stage's last piece: advance the staging cursors behind it
      constexpr int a_first = (nt - BG) * NPG;  // first next-K-tile activation fragment requested in this n-tile
      constexpr bool a_req = nt >= BG && a_first < MT;
      static_assert(nt < NT - 1 || !a_req, "no activation request in the last n-tile");
      using std::integral_constant;
      static_for<MT>([&](auto mt_c) {
        constexpr int mt = decltype(mt_c)::value;
        if constexpr (!(DN_FAT_ABL & 2) || mt == 0) {
          if constexpr (nt < 8) mma_pinned_bf16<true, H16>(acc[nt][mt], w, cur[mt]);
          else mma_pinned_bf16<false, H16>(acc[nt][mt], w, cur[mt]);
        }
        // the gap behind MFMA mt
        if constexpr (nt < NT - 1) {
          if constexpr (mt == 0) {
            if constexpr (nt + 2 < NT) lds_request<(nt + 2) * 1024>(wr[nt % 3], w_cur);
            else lds_request<0>(wa, w_nxt);
          }
          if constexpr (mt == 1 && a_req) lds_request<a_first * 1024>(nxt[a_first], a_nxt);
          if constexpr (dma) {  // (nested: the piece index is only valid where dma holds)
            if constexpr (mt == 2) piece_setup(integral_constant<int, nt - BG>{}, fill);
            if constexpr (mt == 3) piece_go(integral_constant<int, nt - BG>{});
            if constexpr (mt == 3 && dma_last) stage_advance();
          }
          if constexpr (mt == 4 && a_req && NPG == 2) lds_request<(a_first + 1) * 1024>(nxt[a_first + 1], a_nxt);
        } else {
          // last n-tile: once MFMA mt has issued (its operands are read at issue), cur[mt] is free and takes the next
          // K-tile's fragment -- two v_mov_b64 per gap.  All but the last-requested activation fragments are older than this
          // n-tile's weight fragment, which the wait above covered; the last one gets its own counted wait.
          if constexpr (mt == 0) lds_request<1024>(wb, w_nxt);
          if constexpr (dma) {
            if constexpr (mt == 2) piece_setup(integral_constant<int, nt - BG>{}, fill);
            if constexpr (mt == 3) piece_go(integral_constant<int, nt - BG>{});
          }
          if constexpr (mt == MT - 1) lds_wait_all_but<Sched::younger_copy()>();
          copy_after_wait(cur[mt], nxt[mt]);
          if constexpr (mt == 5 && dma_last) stage_advance();
        }
      });
    });
  };

  // ---- HALO form of the K loop.  LDS: a 4-stage ring of weight K-tiles (one per tap of a chunk) and a 2-stage ring of activation
  // chunks, each [16 halo rows ; 256 tile rows] x 64 bytes; tap t reads LDS rows 16 + r - shift_t.  K-tile kt = tap kt % 3 of
  // chunk kt / 3 stages the weight tile of K-tile kt + 3 (6 pieces per wave) and its share of an activation chunk: tap 2 of chunk c
  // the first three row pieces of chunk c + 2 (into chunk c's own slot, which nobody reads any more once every wave is past this
  // K-tile's barrier: the fragments of tap 2 were requested during tap 1), tap 0 of chunk c + 1 the fourth row piece and the wave's
  // 4-row share of the halo.  Chunk c + 2 is first read (the next K-tile's fragments) under tap 2 of chunk c + 1, whose barrier
  // waits for everything but the 6 pieces of the K-tile before it.
  constexpr int A_RING = STAGES * W_BYTES, A_HALO_BYTES = 17 * 1024;
  [[maybe_unused]] uint32_t a_rdh[3] = {0, 0, 0};  // per tap: this lane's fragment address in slot 0 of the activation ring (+ mt * 1024)
  [[maybe_unused]] auto ktile_h = [&](int slot, int aslot, auto tap_c, auto stw_c, auto sta_c, auto sync_c) {
    constexpr int TAP = decltype(tap_c)::value, SYNC = decltype(sync_c)::value;
    constexpr bool STW = decltype(stw_c)::value, STA = decltype(sta_c)::value;
    constexpr int NA = !STA ? 0 : TAP == 2 ? 3 : TAP == 0 ? 2 : 0;  // activation pieces this K-tile stages
    constexpr int PERH = (STW ? WPW : 0) + NA;
    static_assert(STW || NA == 0, "activation pieces ride behind the weight pieces");
    const int nslot = slot == STAGES - 1 ? 0 : slot + 1;
    const uint32_t w_cur = w_rd + slot * W_BYTES, w_nxt = w_rd + nslot * W_BYTES;
    const uint32_t a_nxt = a_rdh[(TAP + 1) % 3] + (TAP == 2 ? (aslot ^ 1) : aslot) * A_HALO_BYTES;
    const int fill = slot == 0 ? STAGES - 1 : slot - 1;       // weights of K-tile kt + 3: the slot K-tile kt - 1 lived in
    const int afill = TAP == 2 ? aslot : (aslot ^ 1);          // chunk c + 2 -> chunk c's slot; (tap 0 of chunk c) chunk c + 1 -> the other one
    auto h_setup = [&](auto i_c) {
      constexpr int i = decltype(i_c)::value;
      if constexpr (i < WPW) {
        int pc = wave + WAVES * i;
        pc = pc < pc_max ? pc : pc_max;
        piece_base = glds_set_m0_base(lds_base + fill * W_BYTES + pc * 1024, tap_wbase + pc * piece_stride);
      } else {
        constexpr int j = i - WPW + (TAP == 0 ? 3 : 0);  // 0..3: the wave's row pieces, 4: its share of the halo
        if constexpr (j < 4) glds_set_m0(lds_base + A_RING + afill * A_HALO_BYTES + 1024 + (wave * APW + j) * 1024);
        else glds_set_m0(lds_base + A_RING + afill * A_HALO_BYTES + wave * 256);
      }
    };
    auto h_go = [&](auto i_c) {
      constexpr int i = decltype(i_c)::value;
      if constexpr (i < WPW) {
        glds_go_s(w_voff, piece_base);
      } else {
        constexpr int j = i - WPW + (TAP == 0 ? 3 : 0);
        if constexpr (j < 4) glds_go(a_ptr[j]);
        else glds_go_lanes16(halo_ptr);
      }
    };
    auto h_advance = [&]() {  // behind the K-tile's last piece
      if constexpr (TAP == 2) { tap_wbase = tap_wbase0; w_voff += w_step; }
      else tap_wbase += tap_wstride;
      if constexpr (TAP == 0 && NA > 0) {
#pragma unroll
        for (int i = 0; i < APW; ++i) a_ptr[i] += tap_akinc;
        halo_ptr += halo_inc;
      }
    };
    static_for<NT>([&](auto nt_c) {
      constexpr int nt = decltype(nt_c)::value;
      if constexpr (nt == BG) pipe_sync<SYNC>();
      u32x4& w = [&]() -> u32x4& {
        if constexpr (nt == 0) return wa;
        else if constexpr (nt == 1) return wb;
        else return wr[(nt - 2) % 3];
      }();
      lds_wait<Sched::younger_w(nt)>(w);
      constexpr bool dma = nt >= BG && nt - BG < PERH;
      constexpr bool dma_last = dma && nt - BG == PERH - 1;
      static_assert(!dma || nt < NT - 1, "no piece in the last n-tile");
      constexpr int a_first = (nt - BG) * NPG;
      constexpr bool a_req = nt >= BG && a_first < MT;
      using std::integral_constant;
      static_for<MT>([&](auto mt_c) {
        constexpr int mt = decltype(mt_c)::value;
        if constexpr (nt < 8) mma_pinned_bf16<true, H16>(acc[nt][mt], w, cur[mt]);
        else mma_pinned_bf16<false, H16>(acc[nt][mt], w, cur[mt]);
        if constexpr (nt < NT - 1) {
          if constexpr (mt == 0) {
            if constexpr (nt + 2 < NT) lds_request<(nt + 2) * 1024>(wr[nt % 3], w_cur);
            else lds_request<0>(wa, w_nxt);
          }
          if constexpr (mt == 1 && a_req) lds_request<a_first * 1024>(nxt[a_first], a_nxt);
          if constexpr (dma) {
            if constexpr (mt == 2) h_setup(integral_constant<int, nt - BG>{});
            if constexpr (mt == 3) h_go(integral_constant<int, nt - BG>{});
            if constexpr (mt == 4 && dma_last) h_advance();
          }
        } else {
          if constexpr (mt == 0) lds_request<1024>(wb, w_nxt);
          if constexpr (mt == MT - 1) lds_wait_all_but<Sched::younger_copy()>();
          copy_after_wait(cur[mt], nxt[mt]);
        }
      });
    });
  };

#ifdef DN_FAT_STAMPS  // diagnostic build only (tools/fat_clock.py): in-kernel clock and K-loop cycles
  const uint64_t dbg_c0 = __builtin_readcyclecounter(), dbg_r0 = __builtin_amdgcn_s_memrealtime();
  uint64_t dbg_c1 = 0;
#endif
  if constexpr (taps_inner) setup_taps(); else setup_term(0);
  // A tile can share the staged rows between the taps if no sequence starts INSIDE it (a frame in front of a sequence must read as
  // zero for the later taps and as the previous sequence's data for its own row): the tile's rows end with its first row's sequence,
  // or what follows lies beyond M.  Other tiles of the launch run the shifted-copies loop below: same K order, same bits.
  bool shared_rows = false;
  if constexpr (HALO) {
    const int t_first = m0 % p.T;
    shared_rows = t_first + BMF <= p.T || m0 + (p.T - t_first) >= p.M;
  }
  if constexpr (HALO) {
  if (shared_rows) {
    using std::integral_constant;
    const int chunks = p.K / KT;  // >= 3 (launch_fat)
    {
      const int tshift[3] = {(int)p.terms[0].shift, (int)p.terms[1].shift, (int)p.terms[2].shift};
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int row = 16 + wm * RPW + frow - tshift[t];
        a_rdh[t] = lds_base + A_RING + row * ROWB2 + ((fq ^ (((row >> 3) & 1) << 1)) << 4);
      }
    }
    // prologue: chunk 0 (whole), the weight tiles of K-tiles 0..2, the first three row pieces of chunk 1
    auto stage_rows = [&](int aslot, int first, int last) {
      for (int j = first; j <= last; ++j) {
        if (j < 4) { glds_set_m0(lds_base + A_RING + aslot * A_HALO_BYTES + 1024 + (wave * APW + j) * 1024); asm volatile("s_nop 0"); glds_go(a_ptr[j]); }
        else { glds_set_m0(lds_base + A_RING + aslot * A_HALO_BYTES + wave * 256); asm volatile("s_nop 0"); glds_go_lanes16(halo_ptr); }
      }
    };
    stage_rows(0, 0, 4);
#pragma unroll
    for (int i = 0; i < APW; ++i) a_ptr[i] += tap_akinc;
    halo_ptr += halo_inc;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      static_for<WPW>([&](auto i_c) {
        int pc = wave + WAVES * decltype(i_c)::value;
        pc = pc < pc_max ? pc : pc_max;
        piece_base = glds_set_m0_base(lds_base + t * W_BYTES + pc * 1024, tap_wbase + pc * piece_stride);
        asm volatile("s_nop 0");
        glds_go_s(w_voff, piece_base);
      });
      if (t == 2) { tap_wbase = tap_wbase0; w_voff += w_step; } else tap_wbase += tap_wstride;
    }
    stage_rows(1, 0, 2);
    pipe_sync<2 * WPW + 3>();  // chunk 0 and the first weight tile have landed (two weight tiles and three row pieces may be in flight)
#ifdef DN_FAT_STAMPS
    dbg_c1 = __builtin_readcyclecounter();
#endif
    static_for<MT>([&](auto mt_c) { lds_request<decltype(mt_c)::value * 1024>(cur[decltype(mt_c)::value], a_rdh[0]); });
    lds_request<0>(wa, w_rd);
    lds_request<1024>(wb, w_rd);
    int slot = 0, aslot = 0;
    auto next_slot = [&]() { slot = slot == STAGES - 1 ? 0 : slot + 1; };
    using T_ = integral_constant<bool, true>;
    using F_ = integral_constant<bool, false>;
    for (int c = 0; c + 2 < chunks; ++c) {
      ktile_h(slot, aslot, integral_constant<int, 0>{}, T_{}, T_{}, integral_constant<int, WPW + 3>{}); next_slot();
      ktile_h(slot, aslot, integral_constant<int, 1>{}, T_{}, T_{}, integral_constant<int, WPW + 2>{}); next_slot();
      ktile_h(slot, aslot, integral_constant<int, 2>{}, T_{}, T_{}, integral_constant<int, WPW>{}); next_slot();
      aslot ^= 1;
    }
    // chunk C - 2: its last tap has no activation chunk left to stage; chunk C - 1: nothing left at all
    ktile_h(slot, aslot, integral_constant<int, 0>{}, T_{}, T_{}, integral_constant<int, WPW + 3>{}); next_slot();
    ktile_h(slot, aslot, integral_constant<int, 1>{}, T_{}, T_{}, integral_constant<int, WPW + 2>{}); next_slot();
    ktile_h(slot, aslot, integral_constant<int, 2>{}, T_{}, F_{}, integral_constant<int, WPW>{}); next_slot();
    aslot ^= 1;
    ktile_h(slot, aslot, integral_constant<int, 0>{}, F_{}, F_{}, integral_constant<int, WPW>{}); next_slot();
    ktile_h(slot, aslot, integral_constant<int, 1>{}, F_{}, F_{}, integral_constant<int, 0>{}); next_slot();
    ktile_h(slot, aslot, integral_constant<int, 2>{}, F_{}, F_{}, integral_constant<int, 0>{});
  }
  }
  if (!shared_rows) {
#pragma unroll
  for (int st = 0; st < STAGES - 1; ++st)
    if (st < nkt) stage(st);
  if (nkt > 2) pipe_sync<2 * PER>(); else if (nkt > 1) pipe_sync<PER>(); else pipe_sync<0>();
#ifdef DN_FAT_STAMPS
  dbg_c1 = __builtin_readcyclecounter();
#endif
  static_for<MT>([&](auto mt_c) { lds_request<decltype(mt_c)::value * 1024>(cur[decltype(mt_c)::value], a_rd); });
  lds_request<0>(wa, w_rd);
  lds_request<1024>(wb, w_rd);
  {
    using std::integral_constant;
    int slot = 0;
    auto next_slot = [&]() { slot = slot == STAGES - 1 ? 0 : slot + 1; };
    for (int kt = 0; kt + 3 < nkt; ++kt) {  // tiles kt+1, kt+2 in flight behind the one being waited for
      ktile(slot, integral_constant<bool, true>{}, integral_constant<int, PER>{});
      next_slot();
    }
    if (nkt >= 3) { ktile(slot, integral_constant<bool, false>{}, integral_constant<int, PER>{}); next_slot(); }
    if (nkt >= 2) { ktile(slot, integral_constant<bool, false>{}, integral_constant<int, 0>{}); next_slot(); }
    ktile(slot, integral_constant<bool, false>{}, integral_constant<int, 0>{});
  }
  }  // !shared_rows
  // The last K-tile still issued its cross-tile requests (stale slot, values unused).  To the compiler those registers
  // are dead the moment they are requested, so it would hand them to epilogue temporaries while the LDS data is still
  // in flight -- and the late return would overwrite them.  Drain the LDS queue with every such register as an operand.
  if constexpr (MT == 8) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(wa), "+v"(wb), "+v"(nxt[0]), "+v"(nxt[1]), "+v"(nxt[2]), "+v"(nxt[3]), "+v"(nxt[MT - 4]), "+v"(nxt[MT - 3]),
                   "+v"(nxt[MT - 2]), "+v"(nxt[MT - 1])
                 :
                 : "memory");
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wa), "+v"(wb), "+v"(nxt[0]), "+v"(nxt[1]), "+v"(nxt[2]), "+v"(nxt[3]) : : "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last MFMA results land before the epilogue reads them
#ifdef DN_FAT_STAMPS
  const uint64_t dbg_c2 = __builtin_readcyclecounter();
#endif
  __syncthreads();
#ifdef DN_FAT_STAMPS
  if ((p.pad_ & (1 << 20)) && tid == 0) {  // stamps go to a buffer of their own (pos_table is unused by this epilogue)
    uint64_t* d = reinterpret_cast<uint64_t*>(const_cast<float*>(p.pos_table)) + 4 * (blockIdx.x + gridDim.x * blockIdx.y);
    d[0] = dbg_c1 - dbg_c0; d[1] = dbg_c2 - dbg_c1; d[2] = __builtin_amdgcn_s_memrealtime() - dbg_r0; d[3] = dbg_r0;
  }
#endif

  // ---- epilogue: the wave's (16 MT) x (16 NTW) sub-tile as MT/4 x ceil(NTW / 4) slabs of 64 x 64 (352: the third carries 48 columns).
  // The slab loop is a run-time loop around ONE copy of the epilogue code (inlined per slab, the FiLM epilogue alone made
  // a 58k-instruction kernel); only the accumulator -> LDS copies, which need compile-time register indices, are per slab.
  float* ep = reinterpret_cast<float*>(smem) + wave * (64 * EP_LD);
  const uint32_t ep_lds = lds_base + (wave * (64 * EP_LD) + frow * EP_LD + fq * 4) * 4;  // this lane's accumulator-fragment slot
  constexpr int NHS = (NTW + 3) / 4, MHS = MT / 4;
  // (a split-RMSNorm consumer on this tile fetches its row factors in the epilogue: the accumulator file leaves no room
  //  to carry them across the K loop)
#pragma unroll 1
  for (int sidx = 0; sidx < MHS * NHS; ++sidx) {
    static_for<MHS * NHS>([&](auto s_c) {
      constexpr int MH = decltype(s_c)::value / NHS, NH = decltype(s_c)::value % NHS;
      constexpr int NTS = NTW - 4 * NH < 4 ? NTW - 4 * NH : 4;  // n-tiles in this slab
      if (sidx == MH * NHS + NH) {
        if constexpr (WAVES == 8) {  // straight from the AGPRs: no v_accvgpr_read, no VGPR pressure (128 VGPRs here)
          static_for<4 * NTS>([&](auto i_c) {
            constexpr int mt = decltype(i_c)::value / NTS, nt = decltype(i_c)::value % NTS;
            acc_to_lds<(mt * 16 * EP_LD + nt * 16) * 4>(ep_lds, acc[NH * 4 + nt][MH * 4 + mt]);
          });
        } else {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTS; ++nt)
              *reinterpret_cast<f32x4*>(ep + (mt * 16 + frow) * EP_LD + nt * 16 + fq * 4) = acc[NH * 4 + nt][MH * 4 + mt];
        }
      }
    });
    const int mh = sidx / NHS, nh = sidx - mh * NHS;
    const int nts = NTW - 4 * nh < 4 ? NTW - 4 * nh : 4;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int m_slab = m0 + wm * RPW + mh * 64, n_slab = n0 + wn * (16 * NTW) + nh * 64;
    if constexpr (WAVES == 8)  // opaque: keeps the per-row address arithmetic inside the loop (hoisted, it overflows the 128 VGPRs)
      asm volatile("" : "+s"(m_slab), "+s"(n_slab));
    wave_epilogue<EPI, std::is_same<E, BF16X3>::value, std::is_same<E, F16>::value>(p, ep, m_slab, n_slab, g, lane, nts * 16, -1.f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // slab reads done before the next slab overwrites it
  }
}

// In-chain launch timing (dn_profile_start / dn_profile_stop): HIP events recorded on the launch stream
// around every dn_conv_gemm whose tag (bits 8..15 of DnGemmParams.pad_) matches.  Eager launches only.
// (struct LaunchProfile / g_prof: common.h -- wgrad_tn.hip times its launches the same way)

template <typename E, int EPI, int BM, int STAGES>
static int launch_tile(const DnGemmParams& p, hipStream_t s) {
  constexpr int ring = STAGES * (W_TILE_BYTES + BM * ROWB), slabs = (BM / 32) * 64 * 68 * 4;
  constexpr int lds = ring > slabs ? ring : slabs;  // K-loop ring, reused as the epilogue's transpose slabs
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_kernel<E, EPI, BM, STAGES>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  const int np = p.N * (EPI == DN_EPI_GEGLU ? 2 : 1);
  dim3 grid(((p.M + BM - 1) / BM) * ((np + BN - 1) / BN), p.groups);
  const bool timed = g_prof.cap > 0 && ((p.pad_ >> 8) & 0xff) == g_prof.tag && g_prof.n < g_prof.cap;
  if (timed) (void)hipEventRecord(g_prof.ev[2 * g_prof.n], s);
  hipLaunchKernelGGL((conv_gemm_kernel<E, EPI, BM, STAGES>), grid, dim3(BM * 2), lds, s, p);
  if (timed) (void)hipEventRecord(g_prof.ev[2 * g_prof.n++ + 1], s);
  DN_CHECK_LAUNCH("dn_conv_gemm");
  return DN_OK;
}

// Are the terms the taps of one causal conv, to be run innermost in K by the two tiles with 32-deep K-tiles (256 x 352, 256 x 256)?
// Same activation tensor and layout, non-negative shifts and weight addresses in arithmetic progression.  That order is those
// tiles' default: the activation panel crosses the fabric once instead of once per tap.  It changes the fp32 summation order, and
// with it the last bits, relative to the 128-byte-K-tile variants (term-outer): a batch large enough to route to these tiles and
// a smaller one do not agree to the last bit; equal-size shards do.  DN_TAPS_INNER=0 (or DN_FAT_TAPS_INNER=0, the older name)
// or bit 23 of pad_ restores term-outer everywhere (bit 22 forces tap-inner).
static inline bool taps_route_by_shape() { return option_or(OPT_TAPS_INNER, 1) == 2; }  // dn_set_option("taps_inner", 2)
static inline bool terms_are_taps(const DnGemmParams& p) {
  const bool env_taps = option_or(OPT_TAPS_INNER, 1) != 0;
  bool taps = p.n_terms >= 2 && p.n_terms <= 4 && !((p.pad_ >> 23) & 1) && (env_taps || ((p.pad_ >> 22) & 1));
  for (int i = 0; i < p.n_terms; ++i) taps = taps && p.terms[i].layout == p.terms[0].layout && p.terms[i].shift >= 0;
  for (int i = 1; i < p.n_terms && taps; ++i) {
    const DnGemmTerm &a = p.terms[i], &b = p.terms[i - 1], &t0 = p.terms[0], &t1 = p.terms[1];
    taps = a.A == t0.A && a.lda == t0.lda && a.a_gstride == t0.a_gstride && a.w_gstride == t0.w_gstride &&
           a.shift_by_group == t0.shift_by_group && b.shift - a.shift == t0.shift - t1.shift && t0.shift >= t1.shift &&
           (intptr_t)a.W - (intptr_t)b.W == (intptr_t)t1.W - (intptr_t)t0.W;
  }
  return taps;
}

template <typename E, int EPI, bool TAPS, int BNB = 256, bool HALO = false>
static void launch_big_variant(const DnGemmParams& p, dim3 grid, int lds, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_big_kernel<E, EPI, TAPS, BNB, HALO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_gemm_big_kernel<E, EPI, TAPS, BNB, HALO>), grid, dim3(512), lds, s, p);
}

// Shared staging of the taps' rows on the 256 x 256 tile: three taps with shifts (2d, d, 0) and at least two K-chunks; per group
// (dilation = shift << group) and per tile the kernel falls back to shifted copies (halo over 128 rows, a sequence start inside the
// tile).  DN_BIG_HALO=0 or bit 21 of pad_ switch it off.
static inline bool big_taps_share_rows(const DnGemmParams& p, int kt_elems) {
  static const bool env_off = getenv("DN_BIG_HALO") && atoi(getenv("DN_BIG_HALO")) == 0;
  if (env_off || ((p.pad_ >> 21) & 1) || p.n_terms != 3 || p.K / kt_elems < 2) return false;
  const DnGemmTerm &t0 = p.terms[0], &t1 = p.terms[1], &t2 = p.terms[2];
  return t2.shift == 0 && t1.shift >= 1 && t0.shift == 2 * t1.shift && (t0.shift_by_group || 2 * t1.shift <= 128);
}

template <typename E, int EPI, int BNB = 256>
static int launch_big(const DnGemmParams& p, hipStream_t s) {
  constexpr int ring = 4 * (BNB + 256) * ROWB2, slabs = 8 * 64 * EP_LD * 4;
  constexpr int lds = ring > slabs ? ring : slabs;
  const int np = p.N * (EPI == DN_EPI_GEGLU ? 2 : 1);
  dim3 grid(((p.M + 255) / 256) * ((np + BNB - 1) / BNB), p.groups);
  const bool timed = g_prof.cap > 0 && ((p.pad_ >> 8) & 0xff) == g_prof.tag && g_prof.n < g_prof.cap;
  if (timed) (void)hipEventRecord(g_prof.ev[2 * g_prof.n], s);
  bool tapped = false;
  if constexpr ((EPI == DN_EPI_BIAS || EPI == DN_EPI_FILM_GATE) && !std::is_same<E, BF16X3>::value && BNB == 256) {  // the epilogues a causal conv has (CausalConv1d + bias; the WaveNet block); split operands pair K-tiles along a row: term-outer only
    if (terms_are_taps(p)) {
      if (big_taps_share_rows(p, ROWB2 / Elem<E>::bytes)) launch_big_variant<E, EPI, true, 256, true>(p, grid, lds, s);
      else launch_big_variant<E, EPI, true>(p, grid, lds, s);
      tapped = true;
    }
  }
  if (!tapped) launch_big_variant<E, EPI, false, BNB>(p, grid, lds, s);
  if (timed) (void)hipEventRecord(g_prof.ev[2 * g_prof.n++ + 1], s);
  DN_CHECK_LAUNCH("dn_conv_gemm");
  return DN_OK;
}

template <typename E, int EPI>
static int launch_mid2(const DnGemmParams& p, hipStream_t s) {
  constexpr int ring = 3 * (BNM + 256) * ROWB2, slabs = 4 * 64 * EP_LD * 4;
  constexpr int lds = ring > slabs ? ring : slabs;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_mid2_kernel<E, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  const int np = p.N * (EPI == DN_EPI_GEGLU ? 2 : 1);
  dim3 grid(((p.M + 255) / 256) * ((np + BNM - 1) / BNM), p.groups);
  const bool timed = g_prof.cap > 0 && ((p.pad_ >> 8) & 0xff) == g_prof.tag && g_prof.n < g_prof.cap;
  if (timed) (void)hipEventRecord(g_prof.ev[2 * g_prof.n], s);
  hipLaunchKernelGGL((conv_gemm_mid2_kernel<E, EPI>), grid, dim3(256), lds, s, p);
  if (timed) (void)hipEventRecord(g_prof.ev[2 * g_prof.n++ + 1], s);
  DN_CHECK_LAUNCH("dn_conv_gemm (256 x 128 tile, two workgroups per CU)");
  return DN_OK;
}

template <typename E, int EPI>
static int launch_row(const DnGemmParams& p, hipStream_t s) {
  constexpr int ring = 2 * (512 + 64) * ROWB, slabs = 8 * 64 * EP_LD * 4 + 64 * 8 * 4;
  constexpr int lds = ring > slabs ? ring : slabs;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_row_kernel<E, EPI>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  dim3 grid((p.M + 63) / 64, p.groups);
  hipLaunchKernelGGL((conv_gemm_row_kernel<E, EPI>), grid, dim3(512), lds, s, p);
  DN_CHECK_LAUNCH("dn_conv_gemm (row tile)");
  return DN_OK;
}

template <typename E, int EPI, bool TAPS_INNER, int NTW, int WAVES, bool HALO = false>
static void launch_fat_variant(const DnGemmParams& p, dim3 grid, int lds, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_fat_kernel<E, EPI, TAPS_INNER, NTW, WAVES, HALO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_gemm_fat_kernel<E, EPI, TAPS_INNER, NTW, WAVES, HALO>), grid, dim3(64 * WAVES), lds, s, p);
}

// Can the three taps share ONE staged copy of the activation rows (the HALO form of the 256 x 352 tile)?  Taps of one causal conv
// with shifts (2d, d, 0), d <= 8; at least three K-chunks.  The kernel decides per tile: one with a sequence start inside runs the
// shifted-copies loop (with sequences a multiple of 256 frames long no tile has one).  DN_FAT_HALO=0 or bit 21 of pad_ switch it
// off (A/B timing, tests).
static inline bool taps_share_rows(const DnGemmParams& p) {
  static const bool env_off = getenv("DN_FAT_HALO") && atoi(getenv("DN_FAT_HALO")) == 0;
  if (env_off || ((p.pad_ >> 21) & 1) || p.n_terms != 3 || p.K < 96) return false;
  const DnGemmTerm &t0 = p.terms[0], &t1 = p.terms[1], &t2 = p.terms[2];
  return !t0.shift_by_group && t2.shift == 0 && t1.shift >= 1 && t1.shift <= 8 && t0.shift == 2 * t1.shift;
}

// NTW = 11: the 256 x 352 tile (caller guarantees N % 352 == 0, BIAS epilogue); NTW = 8: the 256 x 256 tile, with one
// (WAVES = 4) or two (WAVES = 8) waves per SIMD.
template <typename E, int EPI, int NTW, int WAVES = 4>
static int launch_fat(const DnGemmParams& p, hipStream_t s) {
  constexpr int BNF = 32 * NTW;
  constexpr int ring = 4 * (BNF + 256) * ROWB2, slabs = WAVES * 64 * EP_LD * 4;
  constexpr int lds = ring > slabs ? ring : slabs;
  const int np = p.N * (EPI == DN_EPI_GEGLU ? 2 : 1);
  dim3 grid(((p.M + 255) / 256) * ((np + BNF - 1) / BNF), p.groups);
  const bool timed = g_prof.cap > 0 && ((p.pad_ >> 8) & 0xff) == g_prof.tag && g_prof.n < g_prof.cap;
  const bool taps = terms_are_taps(p);  // tap-inner K order for the taps of one causal conv (see the kernel and terms_are_taps)
  if (timed) (void)hipEventRecord(g_prof.ev[2 * g_prof.n], s);
  bool shared = false;
  if constexpr (NTW == 11 && WAVES == 4 && EPI == DN_EPI_BIAS) {
    if (taps && taps_share_rows(p)) {
      launch_fat_variant<E, EPI, true, NTW, WAVES, true>(p, grid, lds, s);
      shared = true;
    }
  }
  if (shared) {}
  else if (taps) launch_fat_variant<E, EPI, true, NTW, WAVES>(p, grid, lds, s);
  else launch_fat_variant<E, EPI, false, NTW, WAVES>(p, grid, lds, s);
  if (timed) (void)hipEventRecord(g_prof.ev[2 * g_prof.n++ + 1], s);
  DN_CHECK_LAUNCH("dn_conv_gemm (one-wave-per-SIMD tile)");
  return DN_OK;
}

// Tile variant forced for this call: DN_GEMM_TILE (process-wide) or bits 16..19 of pad_ (per call; tests); 0 = choose by shape.
static inline int forced_tile(const DnGemmParams& p) {
  static const int env_tile = getenv("DN_GEMM_TILE") ? atoi(getenv("DN_GEMM_TILE")) : 0;
  return ((p.pad_ >> 16) & 15) ? ((p.pad_ >> 16) & 15) : env_tile;
}

// Does this contraction run on the 256 x 352 one-wave-per-SIMD tile?  bf16, BIAS or GEGLU, packed columns a multiple of
// 352 and at least ~half a chip of tiles (a workgroup owns a CU's whole LDS; measured +2.5 % per denoising step on half
// batches, +5 % on whole ones, against the 256 x 256 tile on the FFN conv).  Chosen by itself only for the long-K BIAS
// contractions; on the GEGLU projection (K = 512: 16 K-tiles) it measured level with the 256 x 256 tile (67.6 vs 66.1
// us), so there it runs only when forced.
static inline bool routes_to_352(const DnGemmParams& p) {
  if (!dn_is16(p.dtype) || (p.epilogue != DN_EPI_BIAS && p.epilogue != DN_EPI_GEGLU)) return false;
  if ((p.epilogue == DN_EPI_RESADD || p.epilogue == DN_EPI_POSEMB) && p.norm_out && !p.norm_split) return false;
  const int force = forced_tile(p);
  const int npk = p.N * (p.epilogue == DN_EPI_GEGLU ? 2 : 1);
  if (npk % 352 != 0) return false;
  if (force == 4) return true;
  if (force != 0 || p.epilogue != DN_EPI_BIAS) return false;
  // One workgroup per CU on both 256-row tiles, so what decides is how much of the chip a launch fills: the 352-wide tile is
  // ~1.3x the 256-wide one per flop on a full chip (1.33 vs 0.95-1.0 PFLOP/s on the FFN conv) but has 1/1.375 of its tiles.  A
  // half-batch launch of the two-stream sampling chain (bit 7 of pad_: an identical twin runs beside it) counts double: 128 + 128
  // workgroups fill the chip.  Alone, a 128-tile launch leaves half the chip idle: the training step's FFN conv at M = 8192 takes
  // 138 us on this tile against 98 us on 192 tiles of 256 x 256.
  const long twin = (p.pad_ & 128) ? 2 : 1;
  const long tiles_fat = (long)((p.M + 255) / 256) * (npk / 352) * p.groups * twin;
  const long tiles_big = (long)((p.M + 255) / 256) * ((npk + 255) / 256) * p.groups * twin;
  auto fill = [](long tiles) { const double r = (double)tiles / 256.0; return r / ceil(r); };
  static const bool old_rule = getenv("DN_352_ROUTE") && atoi(getenv("DN_352_ROUTE")) == 0;  // A/B timing: at least 100 tiles, whatever they fill
  if (old_rule) return tiles_fat / twin >= 100;
  return tiles_fat >= 100 && 1.3 * fill(tiles_fat) >= fill(tiles_big);
}

// The tile variant a contraction runs on: 1 = 128 x 128, 2 = 256 x 128, 3 = 256 x 256, 4 = 256 x 352, 5 = whole-row (fused norm),
// 6 / 7 = the forced-only hand-scheduled 256 x 256 forms; -1 = K-blocked operands with a tile forced that does not take them.
static inline int choose_tile(const DnGemmParams& p) {
  const bool bf = dn_is16(p.dtype);
  if ((p.epilogue == DN_EPI_RESADD || p.epilogue == DN_EPI_POSEMB) && p.norm_out && !p.norm_split && p.dtype != DN_BF16X3) return 5;
  const int force = forced_tile(p);
  bool has_ldw = false;
  for (int i = 0; i < p.n_terms; ++i) has_ldw = has_ldw || p.terms[i].ldw != 0;
  // A handful of rows against a huge fp32 weight (the conditioning projections: B rows x [2048 -> 57 k], 470 MB of weights): the
  // launch is a weight stream, and what decides is how the rows of the weight are fetched -- the 128-byte K-tiles of the 128 x 128
  // tile take whole cache lines per row (64-byte K-tiles: half lines from rows 8 KiB apart), two workgroups per CU keep twice
  // the loads in flight.  Measured at B = 16 (tools/cond_gemm_bench.py): 282 us on that tile against 503-569 us on the others.
  if (p.dtype == DN_F32 && p.M <= 128 && force == 0 && (long)p.N * p.K * p.groups >= (8L << 20)) return 1;
  if (has_ldw && p.dtype != DN_BF16X3) {  // a weight row stride other than K: the 128x128 / 256x128 / 256x256 tiles (row-major operands)
    const int npw = p.N * (p.epilogue == DN_EPI_GEGLU ? 2 : 1);
    const long tb = (long)((p.M + 255) / 256) * ((npw + 255) / 256) * p.groups, ts = (long)((p.M + 127) / 128) * ((npw + BN - 1) / BN) * p.groups;
    if (force >= 1 && force <= 3) return force;
    return tb >= 256 ? 3 : ts >= 512 ? 2 : 1;
  }
  if (p.dtype == DN_BF16X3) {  // split operands run on the two 128-byte-K-tile kernels (row-major operands only)
    for (int i = 0; i < p.n_terms; ++i)
      if (p.terms[i].layout != 0) return -1;
    if ((force >= 1 && force <= 3) || (force == 8 && p.epilogue == DN_EPI_BIAS)) return force;
    const int npx = p.N * (p.epilogue == DN_EPI_GEGLU ? 2 : 1);
    const long mt2 = (p.M + 255) / 256, t_big = mt2 * ((npx + 255) / 256) * p.groups, t_mid = mt2 * ((npx + BN - 1) / BN) * p.groups,
               t_small = (long)((p.M + 127) / 128) * ((npx + BN - 1) / BN) * p.groups;
    auto fill2 = [](long tiles, int per_cu) { const double r = (double)tiles / (256.0 * per_cu); return r / ceil(r); };
    // three MFMAs per product against twice the staged bytes: the 256 x 256 tile is the one that stays matrix-bound (its fill
    // path needs 21 B/cycle per CU; 256 x 128 needs 31, two 128 x 128 workgroups 43 of the ~30 available)
    const double sb = 1.00 * fill2(t_big, 1), sm = 0.80 * fill2(t_mid, 1), ss = 0.70 * fill2(t_small, 2);
    // The 256 x 192 tile (tile 8, BIAS epilogue; N = 1408 -> 512 workgroups = two full rounds instead of 1.5) is forced-only
    // (DN_X3_192=1 lets the score pick it): isolated, the FFN conv + q/kv pair of a layer drops 552 -> 513 us at [32,512], but
    // in the two-stream chain each half-batch launch is then exactly one round and the halves stop filling each other's ragged
    // rounds: 77.9 vs 80.4 steps/s, three alternating same-box runs.
    static const bool use192 = getenv("DN_X3_192") && atoi(getenv("DN_X3_192")) != 0;
    if (p.epilogue == DN_EPI_BIAS && force == 0 && use192) {
      const int c192 = (npx + 191) / 192;
      const double s192 = 0.97 * fill2(mt2 * c192 * p.groups, 1) * npx / (192.0 * c192) / (npx / (256.0 * ((npx + 255) / 256)));
      if (s192 > sb && s192 > sm && s192 > ss) return 8;
    }
    if (sb >= sm && sb >= ss) return 3;
    return sm > ss ? 2 : 1;
  }
  if (routes_to_352(p)) return 4;
  // DN_TAPS_INNER=2: the taps of a causal conv run on the 64-byte-K-tile kernels (tap-inner order, shared rows) WHATEVER M is, so a
  // batch and its shards sum every output in the same order -- bit-for-bit sharding invariance at the fast order's speed on
  // large batches (small ones pay for 256-row tiles); the sharded driver's default (normalize.py).
  if (force == 0 && (p.epilogue == DN_EPI_BIAS || p.epilogue == DN_EPI_FILM_GATE) && taps_route_by_shape() && terms_are_taps(p)) return 3;
  bool kblocked = false;
  for (int i = 0; i < p.n_terms; ++i) kblocked = kblocked || p.terms[i].layout != 0;
  const bool mid2_ok = bf && p.epilogue != DN_EPI_RESADD && p.epilogue != DN_EPI_POSEMB;  // tile 9: 256 x 128, two workgroups per CU (no residual prefetch / split-norm producer on it)
  if (kblocked) return bf && force == 8 && p.epilogue == DN_EPI_BIAS ? 8 : mid2_ok && force == 9 ? 9 : bf && (force == 0 || force == 3) ? 3 : -1;  // the other tiles that take them (8: the 192-column form)
  if (bf && force == 8 && (p.epilogue == DN_EPI_BIAS || (p.epilogue == DN_EPI_RESADD && !(p.norm_out && p.norm_split)))) return 8;
  if (mid2_ok && force == 9) return 9;
  if (p.dtype == DN_BF16 && (force == 6 || force == 7)) return force;
  if (force >= 1 && force <= 3) return force;

  const int np = p.N * (p.epilogue == DN_EPI_GEGLU ? 2 : 1);
  const long mt256 = (p.M + 255) / 256, mt128 = (p.M + 127) / 128;
  const long tiles_big = mt256 * ((np + 255) / 256) * p.groups;
  const long tiles_mid = mt256 * ((np + BN - 1) / BN) * p.groups;
  const long tiles_small = mt128 * ((np + BN - 1) / BN) * p.groups;
  // Choose by how evenly the tiles fill the 256 CUs: score = (throughput of the variant on a full chip, relative)
  // x (rounds / ceil(rounds)), rounds = tiles / (CUs x workgroups that share a CU).  The 256 x 256 and 256 x 128
  // rings own a CU's LDS (one workgroup per CU); two 128 x 128 workgroups share one and cover each other's
  // prologue and epilogue, which is what the short-K contractions (K = 512: 8 K-tiles) are made of.
  static const int heur = getenv("DN_GEMM_HEUR") ? atoi(getenv("DN_GEMM_HEUR")) : 1;
  if (heur == 0)  // previous rule, kept for A/B timing
    return tiles_big >= 360 ? 3 : tiles_mid >= 192 ? 2 : 1;
  // The GEGLU projection goes to the 256 x 256 tile as soon as that gives every CU a tile: there the engine feeds it K-blocked
  // operands (whole cache lines), and in the two-stream chain the other half-batch's launches fill its partial last round, which
  // the fill model below cannot see.  Measured at M = 8192 (a half-batch stream of [32,512]): 38.7-40.9 us against 41.6-42.3 on
  // the 128 x 128 tile in isolation, +2.2 % per denoising step in the chain (176.9 vs 173.1 steps/s, two alternating runs).
  if (bf && p.epilogue == DN_EPI_GEGLU && tiles_big >= 256 && heur != 3) return 3;  // DN_GEMM_HEUR=3: scored like the rest (A/B)
  auto fill = [](long tiles, int per_cu) {
    const double rounds = (double)tiles / (256.0 * per_cu);
    return rounds / ceil(rounds);
  };
  // A long-K contraction of several terms (a conv's taps, forward or backward-data) has no prologue / epilogue for a neighbour
  // workgroup to cover: there the smaller tiles are worth 0.72-0.78 of the 256 x 256 tile per output, not 0.9 -- and 0.60-0.68 where the
  // taps share one staged copy of their rows, which only the 256 x 256 tile does.  Measured, bf16, conv k = 3, 2048 wide, M = 12288
  // (tools/bwd_tiles.py): forward 244 us on 384 tiles of 256 x 256 (1.5 rounds, shared rows) against 269 on 768 of 256 x 128 and 309
  // on 1536 of 128 x 128 (3 full rounds each); backward-data (negative shifts: no shared rows) 268 / 259 / 281.  M = 15360: 293
  // against 411.  DN_GEMM_HEUR=4: the short-K factors for every shape (A/B timing).
  const bool multi_long = heur != 4 && (p.epilogue == DN_EPI_BIAS || p.epilogue == DN_EPI_FILM_GATE) && p.n_terms >= 2 && (long)p.K * p.n_terms >= 1024;
  const bool long_taps = multi_long && terms_are_taps(p);
  const bool shares = long_taps && big_taps_share_rows(p, 32) && heur != 5;
  const bool long_k = heur == 5 ? long_taps : multi_long;  // DN_GEMM_HEUR=5: the rule before round 4's recalibration (taps only, no shared-row factor; A/B timing)
  const double s_big = 1.00 * fill(tiles_big, 1), s_mid = (shares ? 0.68 : long_k ? 0.78 : 0.90) * fill(tiles_mid, 1),
               s_small = (shares ? 0.60 : long_k ? 0.72 : 0.92) * fill(tiles_small, 2);
  // The 256 x 192 form of the 256 x 256 kernel (tile 8; BIAS and RESADD epilogues, 2-byte operands, not the taps of a long conv: those
  // share staged rows on the 256-wide form only): widths that are whole multiples of 192 but leave 256-wide tiles a ragged round --
  // the VAE's N = 768 contractions at M = 12288 are 144 tiles of 256 x 256 (0.56 of one round) but 192 of 256 x 192 (0.75 of one,
  // each 3/4 of the work).  Scored only for a launch that has the chip to itself (see g_gemm_twin; option tile_192 = 0: never) and
  // only from K = 768 up: the narrower tile saves MFMAs and weight-panel reads, not the activation panel's LDS traffic nor the
  // prologue and epilogue a short K loop is made of (measured in the training updates: N = 768, K = 768-4096: 75.5 -> 66.2 us;
  // N = 1408, K = 512 at M = 8192, 192 -> 256 workgroups: 77.2 -> 78.8 us).
  if (bf && (p.epilogue == DN_EPI_BIAS || p.epilogue == DN_EPI_RESADD) && !long_taps && !g_gemm_twin && !(p.norm_out && p.norm_split) &&
      (long)p.K * p.n_terms >= 768 && option_or(OPT_TILE_192, 1) != 0) {
    const int c192 = (np + 191) / 192, c256 = (np + 255) / 256;
    const double s_192 = 0.90 * fill(mt256 * c192 * p.groups, 1) * (256.0 * c256) / (192.0 * c192);
    if (s_big >= s_mid && s_big >= s_small && s_192 > s_big) return 8;  // (against the same kernel's 256-wide form only: a pure fill gain)
  }
  if (s_big >= s_mid && s_big >= s_small) return 3;
  return s_mid > s_small ? 2 : 1;
}

// Row-tile band of the tile order (tile_coords) for the 128 x 128 / 256 x 128 / 256 x 256 tiles: 1 unless the weights of one
// group are much larger than an XCD's 4 MB L2 -- then the tiles an XCD runs together (32 CUs x workgroups per CU) should form a
// block that balances activation-panel bytes against weight-panel bytes, band = sqrt(concurrency x weight panel / row panel).
// (The VAE's FFN conv -- 25 MB of weights on 128 x 128 tiles -- re-streamed its weights once per four row tiles: 1.1 GB of fabric
// traffic per launch at 5.3 TB/s.)  DN_GEMM_BAND forces a value for every launch (0 / 1 = column tiles fastest).
static inline int choose_band(const DnGemmParams& p, int tile) {
  static const int env_band = getenv("DN_GEMM_BAND") ? atoi(getenv("DN_GEMM_BAND")) : -1;
  if (env_band >= 0) return env_band < 255 ? env_band : 255;
  const int es = dn_is16(p.dtype) ? 2 : 4;
  const int bm = tile == 1 ? 128 : 256, bn = tile == 3 ? 256 : 128, conc = 32 * (tile == 1 || tile == 9 ? 2 : 1);
  const int np = p.N * (p.epilogue == DN_EPI_GEGLU ? 2 : 1);
  const double w_total = (double)np * p.K * p.n_terms * es;
  if (w_total < 6.0e6) return 1;
  bool same_a = true;
  for (int i = 1; i < p.n_terms; ++i) same_a = same_a && p.terms[i].A == p.terms[0].A;
  const double a_panel = (double)bm * p.K * es * (same_a ? 1 : p.n_terms), w_panel = (double)bn * p.K * p.n_terms * es;
  int band = (int)(sqrt(conc * w_panel / a_panel) + 0.5);
  const int m_tiles = (p.M + bm - 1) / bm;
  band = band < 1 ? 1 : band > m_tiles ? m_tiles : band;
  return band < 255 ? band : 255;
}

template <typename E, int EPI>
static int launch(const DnGemmParams& p0, hipStream_t s) {
  DnGemmParams p = p0;
  const int tile = choose_tile(p);
  static const bool no_res_prefetch = getenv("DN_RES_PREFETCH") && atoi(getenv("DN_RES_PREFETCH")) == 0;  // A/B timing
  if (no_res_prefetch) p.pad_ |= 32;
  if (((tile >= 1 && tile <= 3) || tile == 9) && ((p.pad_ >> 24) & 0xff) == 0)  // bits 24..31 of pad_: a band forced by the caller (tests)
    p.pad_ = (p.pad_ & 0x00ffffff) | (choose_band(p, tile) << 24);
  DN_CHECK_ARG(tile > 0, "dn_conv_gemm: K-blocked operands are taken by the 256 x 352 and 256 x 256 tiles only (bf16; forced tile %d)",
               forced_tile(p));
  if constexpr (std::is_same<E, BF16X3>::value) {  // split operands: the 128-byte-K-tile kernels (both halves in one K-tile) and the 256 x 256 tile (K-tile pairs)
    if constexpr (EPI == DN_EPI_BIAS) {
      if (tile == 8) return launch_big<E, EPI, 192>(p, s);
    }
    if (tile == 3 || tile == 8) return launch_big<E, EPI>(p, s);
    if (tile == 2) return launch_tile<E, EPI, 256, 3>(p, s);
    return launch_tile<E, EPI, 128, 2>(p, s);
  } else {
  if constexpr (EPI == DN_EPI_RESADD || EPI == DN_EPI_POSEMB) {
    if (tile == 5) return launch_row<E, EPI>(p, s);
  }
  if constexpr ((EPI == DN_EPI_BIAS || EPI == DN_EPI_GEGLU) && IsHalf<E>::value) {
    if (tile == 4) return launch_fat<E, EPI, 11>(p, s);
  }
  if constexpr ((EPI == DN_EPI_BIAS || EPI == DN_EPI_RESADD) && IsHalf<E>::value) {
    if (tile == 8) return launch_big<E, EPI, 192>(p, s);  // 256 x 192: widths that are whole multiples of 192 but ragged on 256
  }
  if constexpr (IsHalf<E>::value && EPI != DN_EPI_RESADD && EPI != DN_EPI_POSEMB) {
    if (tile == 9) return launch_mid2<E, EPI>(p, s);
  }
  if constexpr (std::is_same<E, BF16>::value) {  // (the forced-only hand-scheduled 256 x 256 forms: kept for bf16)
    if (tile == 6) return launch_fat<E, EPI, 8>(p, s);
    if (tile == 7) return launch_fat<E, EPI, 8, 8>(p, s);
  }
  if (tile == 3) return launch_big<E, EPI>(p, s);
  if (tile == 2) return launch_tile<E, EPI, 256, 3>(p, s);
  return launch_tile<E, EPI, 128, 2>(p, s);
  }
}

template <typename E>
static int dispatch_epi(const DnGemmParams& p, hipStream_t s) {
  switch (p.epilogue) {
    case DN_EPI_BIAS: return launch<E, DN_EPI_BIAS>(p, s);
    case DN_EPI_SILU: return launch<E, DN_EPI_SILU>(p, s);
    case DN_EPI_RELU: {  // the SILU kernels with the activation switched by pad_ bit 6 (no further instantiations)
      DnGemmParams q = p;
      q.epilogue = DN_EPI_SILU;
      q.pad_ |= 64;
      return launch<E, DN_EPI_SILU>(q, s);
    }
    case DN_EPI_GEGLU: return launch<E, DN_EPI_GEGLU>(p, s);
    case DN_EPI_FILM_GATE: return launch<E, DN_EPI_FILM_GATE>(p, s);
    case DN_EPI_RESADD: return launch<E, DN_EPI_RESADD>(p, s);
    case DN_EPI_POSEMB: return launch<E, DN_EPI_POSEMB>(p, s);
  }
  dn_set_error("dn_conv_gemm: unknown epilogue %d", p.epilogue);
  return DN_EINVAL;
}

// entry points per arithmetic, one translation unit each (gemm_bf16.hip, gemm_f32.hip)
int gemm_dispatch_bf16(const DnGemmParams& p, hipStream_t s);
int gemm_dispatch_f32(const DnGemmParams& p, hipStream_t s);
int gemm_dispatch_x3(const DnGemmParams& p, hipStream_t s);
int gemm_dispatch_f16(const DnGemmParams& p, hipStream_t s);

}  // namespace dn
