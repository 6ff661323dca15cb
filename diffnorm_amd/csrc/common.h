// Device-side helpers shared by the gfx950 kernels of libdiffnorm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/diffnorm_hip.h"

namespace dn {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

struct BF16 {};  // tag types selecting the contraction arithmetic
struct F32 {};
struct BF16X3 {};  // split operands: (hi, lo) bf16 pairs in "split rows" (include/diffnorm_hip.h), three bf16 MFMAs per product
struct F16 {};     // IEEE half operands (v_mfma_f32_16x16x32_f16: the bf16 rate, 11 significand bits, |x| <= 65504); same layouts as BF16

template <typename E> struct Elem;
template <> struct Elem<BF16> { static constexpr int bytes = 2; static constexpr int kDtype = DN_BF16; };
template <> struct Elem<F32> { static constexpr int bytes = 4; static constexpr int kDtype = DN_F32; };
template <> struct Elem<BF16X3> { static constexpr int bytes = 4; static constexpr int kDtype = DN_BF16X3; };
template <> struct Elem<F16> { static constexpr int bytes = 2; static constexpr int kDtype = DN_F16; };
// the two 2-byte operand types share every layout, tile and schedule; they differ in the MFMA opcode and the conversions
template <typename E> struct IsHalf { static constexpr bool value = false; };
template <> struct IsHalf<BF16> { static constexpr bool value = true; };
template <> struct IsHalf<F16> { static constexpr bool value = true; };
__host__ __device__ inline bool dn_is16(int dtype) { return dtype == DN_BF16 || dtype == DN_F16; }

// One "k-step" = 64 bytes of K per row; a fragment is the 16 bytes a lane holds of it:
// lane l -> row (l & 15), 16-byte chunk (l >> 4) of the 64-byte k-step.
//   bf16: 8 consecutive k per lane, one v_mfma_f32_16x16x32_bf16 per k-step (32 k).
//   f32 : 4 consecutive k per lane, four v_mfma_f32_16x16x4_f32 per k-step (16 k); MFMA j consumes
//         element j of every lane, i.e. k-slot q of MFMA j is k = 4q + j.  A and B fragments use the
//         same lane->k map, so the sum over the k-step is exact.
// acc: C[row = 4*(l>>4) + r][col = l & 15] for the A-operand rows / B-operand rows (cols).
template <typename E>
__device__ __forceinline__ void mma_kstep(f32x4& acc, const uint4& a, const uint4& b);

template <>
__device__ __forceinline__ void mma_kstep<BF16>(f32x4& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

template <>
__device__ __forceinline__ void mma_kstep<F16>(f32x4& acc, const uint4& a, const uint4& b) {
  typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}

template <>
__device__ __forceinline__ void mma_kstep<F32>(f32x4& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 v = {(__bf16)lo, (__bf16)hi};  // v_cvt_pk_bf16_f32, round-to-nearest-even, NaN preserved
  return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// IEEE half: v_cvt_pk_f16_f32 (round-to-nearest-even).  A value beyond 65504 becomes inf unless the wave runs with the MODE
// register's FP16_OVFL bit set (f16_saturate(): overflowed fp16 results clamp to +-65504, true infinities stay) -- the DN_F16
// kernels set it at entry, so an operand that leaves the format's range saturates instead of poisoning a contraction with inf - inf.
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
  f16x2 v = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float f16_to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ void f16_saturate() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }
// 16-bit storage by kind: H16 false = bf16, true = IEEE half
template <bool H16>
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) { if constexpr (H16) return pack_f16x2(lo, hi); else return pack_bf16x2(lo, hi); }
template <bool H16>
__device__ __forceinline__ float2 unpack_h2(uint32_t v) {
  if constexpr (H16) {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
    const f16x2 h = __builtin_bit_cast(f16x2, v);
    return make_float2((float)h.x, (float)h.y);
  } else {
    return make_float2(__uint_as_float(v << 16), __uint_as_float(v & 0xffff0000u));
  }
}
// the same with the clamp spelled out (v_med3_f32), for kernels that do not switch the mode bit: the element-wise ones and the
// run-time-dtype store4 below (a handful of stores per thread: the two extra VALU instructions do not show)
__device__ __forceinline__ float f16_clamp(float v) { return __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f); }
__device__ __forceinline__ uint32_t pack_f16x2_sat(float lo, float hi) { return pack_f16x2(f16_clamp(lo), f16_clamp(hi)); }
__device__ __forceinline__ uint16_t to_h16(int dtype, float v) { return (uint16_t)((dtype == DN_F16 ? pack_f16x2_sat(v, 0.f) : pack_bf16x2(v, 0.f)) & 0xffff); }
__device__ __forceinline__ float from_h16(int dtype, uint16_t h) { return dtype == DN_F16 ? f16_to_f32(h) : bf16_to_f32(h); }

// DN_BF16X3 split rows: element `off` of a tensor (row strides multiples of 32) lives as hi at byte (off / 32) * 128 +
// (off % 32) * 2 and lo 64 bytes further on; hi = bf16(x), lo = bf16(x - hi) (the difference is exact in fp32).
__device__ __forceinline__ int64_t split_byte(int64_t off) { return ((off >> 5) << 7) + ((off & 31) << 1); }
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = pack_bf16x2(a, b);
  lo = pack_bf16x2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
// 4 consecutive elements (off a multiple of 4: they stay inside one 32-element group)
__device__ __forceinline__ void store4_split(void* base, int64_t off, float a, float b, float c, float d) {
  uint32_t h0, l0, h1, l1;
  split_pair(a, b, h0, l0);
  split_pair(c, d, h1, l1);
  char* p = reinterpret_cast<char*>(base) + split_byte(off);
  *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(p + 64) = make_uint2(l0, l1);
}
__device__ __forceinline__ void store1_split(void* base, int64_t off, float v) {
  uint32_t h, l;
  split_pair(v, 0.f, h, l);
  char* p = reinterpret_cast<char*>(base) + split_byte(off);
  *reinterpret_cast<uint16_t*>(p) = (uint16_t)h;
  *reinterpret_cast<uint16_t*>(p + 64) = (uint16_t)l;
}
__device__ __forceinline__ float load1_split(const void* base, int64_t off) {
  const char* p = reinterpret_cast<const char*>(base) + split_byte(off);
  return bf16_to_f32(*reinterpret_cast<const uint16_t*>(p)) + bf16_to_f32(*reinterpret_cast<const uint16_t*>(p + 64));
}

// store 4 consecutive fp32 values as out_dtype at element pointer (row base + col)
__device__ __forceinline__ void store4(void* base, int64_t elem_off, int out_dtype, float a, float b, float c, float d) {
  if (out_dtype == DN_BF16X3) {
    store4_split(base, elem_off, a, b, c, d);
  } else if (out_dtype == DN_BF16) {
    uint2 v = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + elem_off) = v;
  } else if (out_dtype == DN_F16) {
    uint2 v = make_uint2(pack_f16x2_sat(a, b), pack_f16x2_sat(c, d));
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + elem_off) = v;
  } else {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + elem_off) = make_float4(a, b, c, d);
  }
}

__device__ __forceinline__ float4 load4(const void* base, int64_t elem_off, int dtype) {
  if (dtype == DN_BF16) {
    uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + elem_off);
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                       __uint_as_float(v.y & 0xffff0000u));
  }
  if (dtype == DN_F16) {
    uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + elem_off);
    const float2 a = unpack_h2<true>(v.x), b = unpack_h2<true>(v.y);
    return make_float4(a.x, a.y, b.x, b.y);
  }
  return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem_off);
}

// Reductions over the four lanes {l, l^16, l^32, l^48} with the gfx950 half/row swaps instead of ds_bpermute:
// v_permlane32_swap(a, b) exchanges lanes 32-63 of a with lanes 0-31 of b, v_permlane16_swap the odd 16-lane rows of a
// with the even rows of b; fed the same value twice they leave {x[l], x[l^32]} resp. {x[l], x[l^16]} in the two results.
__device__ __forceinline__ float quad_xor_max(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ __forceinline__ float quad_xor_sum(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// Epilogue transcendentals.  These run once per output element inside MFMA kernels, so they are built
// from one v_exp_f32 (+ v_rcp_f32) each instead of the libm expansions; absolute error <= ~3e-7, far
// inside the fp32 budget of 1e-3 (checked against the oracle's exact forms in tests/test_hip_ops.py).
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// exact-GELU 0.5 x (1 + erf(x / sqrt 2)) with erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7)
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = fast_rcp(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * __expf(-z * z);
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }
// tanh(h) * sigmoid(h) from one exponential: q = e^-h, sigmoid = 1/(1+q), tanh = (1-q^2)/(1+q^2).
// |h| is clamped to 30 first: beyond it the product is constant to 1e-13.
__device__ __forceinline__ float tanh_sigmoid_gate(float h) {
  const float hc = fminf(fmaxf(h, -30.0f), 30.0f);
  const float q = __expf(-hc);
  const float q2 = q * q;
  return (1.0f - q2) * fast_rcp((1.0f + q2) * (1.0f + q));
}

}  // namespace dn

namespace dn {
// Attention-dropout keep decision, a counter-based hash of (seed, row = (b * heads + h) * T + query, key): the forward and both
// backward kernels regenerate the same mask from the same seed (nothing is stored).  thr = p * 2^32: keep iff hash >= thr.
// XCD-aware work order for a (blocks, heads, batch) grid.  Workgroups are dispatched round-robin over the 8 XCDs in linear id
// order (x fastest), so the row blocks of one (batch, head) -- which all stream the same K / V (or Q / dO) rows -- land on
// different L2s and every one of them fetches those rows through the fabric (measured: 151 MB read per attention launch at
// [32,512] against 48 MB of q/k/v).  Remap: XCD (id & 7) owns a contiguous run of logical ids, block index fastest, so the
// blocks of a (batch, head) are neighbours behind one L2.  Bijective for any grid size.
__device__ __forceinline__ void dn_xcd_block_map(int& bx, int& by, int& bz) {
  const int nx = gridDim.x, ny = gridDim.y, nwg = nx * ny * (int)gridDim.z;
  const int id = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
  const int xcd = id & 7, q = nwg >> 3, r = nwg & 7;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  bx = logical % nx;
  const int t = logical / nx;
  by = t % ny;
  bz = t / ny;
}

__device__ __forceinline__ uint32_t dn_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t dn_drop_row(uint32_t row, uint32_t seed_lo) { return dn_mix32(row ^ seed_lo); }
__device__ __forceinline__ bool dn_drop_keep(uint32_t row_hash, uint32_t key, uint32_t seed_hi, uint32_t thr) {
  return dn_mix32(row_hash + key * 0x9E3779B9U + seed_hi) >= thr;
}
// In-chain launch timing (dn_profile_start / dn_profile_stop, gemm.hip): HIP events recorded on the launch stream around every
// tagged contraction launch.
struct LaunchProfile {
  int tag = 0, cap = 0, n = 0;
  hipEvent_t* ev = nullptr;  // 2*cap events
};
extern LaunchProfile g_prof;  // defined in gemm.hip
// Set by the two-stream sampling loop while it enqueues (or captures) a step: every contraction it launches has an identical
// half-batch twin running beside it, which fills the idle part of a ragged round -- tile choices that even out the rounds of ONE
// launch (the 256 x 192 tile) lose there.  Defined in gemm.hip.
extern thread_local bool g_gemm_twin;

}  // namespace dn

// host-side error plumbing (defined in capi.hip)
void dn_set_error(const char* fmt, ...);

// Run-time options of the library (dn_set_option / dn_get_option, include/diffnorm_hip.h).  Each starts from the environment
// variable of the same meaning, read ONCE when the library first looks (a process-wide default), and is switched afterwards through
// the C ABI only -- no per-launch getenv, no environment mutation by callers.  DN_OPT_UNSET: neither set nor in the environment.
namespace dn {
enum Opt { OPT_TAPS_INNER = 0, OPT_FUSE_NORM, OPT_NO_SPLIT_NORM, OPT_KBLOCK, OPT_WGRAD_STREAM, OPT_WGRAD_TN, OPT_WGRAD_GROUPS, OPT_QKV_192, OPT_MID2, OPT_WGRAD_STAGES, OPT_TILE_192, OPT_FUSED_GEGLU, OPT_ATTN_WAVES8, OPT_COND_STREAM, OPT_WGRAD_K192, OPT_WGRAD_PRIO, OPT_COUNT };
constexpr int DN_OPT_UNSET = -2147483647 - 1;
int option(Opt o);                       // current value or DN_OPT_UNSET
inline int option_or(Opt o, int dflt) { const int v = option(o); return v == DN_OPT_UNSET ? dflt : v; }
}  // namespace dn
#define DN_CHECK_ARG(cond, ...)  \
  do {                           \
    if (!(cond)) {               \
      dn_set_error(__VA_ARGS__); \
      return DN_EINVAL;          \
    }                            \
  } while (0)
#define DN_CHECK_LAUNCH(what)                                             \
  do {                                                                    \
    hipError_t e__ = hipGetLastError();                                   \
    if (e__ != hipSuccess) {                                              \
      dn_set_error("%s: %s", what, hipGetErrorString(e__));               \
      return DN_ELAUNCH;                                                  \
    }                                                                     \
  } while (0)
