// Weight gradient of a causal conv / Linear straight from the ROW-MAJOR operands (autograd of CausalConv1d
// latent_module.py:476-485 and nn.Linear): part[slice][n][tap * rows_w + k] = sum over the slice's frames m of
// dY[m][n] * X_tap[m - shift_tap][k] (zero where the frame index of m is < shift_tap).
//
// The contraction index (frames) is the ROW index of both operands, so neither is K-contiguous as an MFMA fragment wants it; the
// forward kernels' form needs channel-major copies of both (dn_transpose_pad: one pass over dY and one per tap over X, 8 % of a
// training update).  Here a K-tile is 32 frames x 256 columns of each operand staged row-major by LDS-DMA (rows of 512 B, two rows
// per 1 KiB piece) and the fragments come out of LDS through the transposing read ds_read_tr16_b64: a 16-lane group reads a
// 4 (frames) x 16 (columns) block and every lane receives one column's 4 frames; two reads (frames r and r + 16) make the 8 k-values
// of a 16x16x32 fragment -- the same permutation of the 32 frames in both operands.  The 32-byte granules of a row are XOR-swizzled
// with the row index (on the DMA's per-lane source address and on the read address), so the 16 rows one read touches fall into 16
// different granules.  MFMA A operand = X^T (so a lane ends up with 4 consecutive k of one output row n: float4 stores into
// [n][k]), B operand = dY^T.  Tile 256 (k) x 256 (n), 8 waves of 64 x 128, the 256 x 256 forward kernel's pipeline: 4-stage ring
// of 32 KiB K-tiles, two wave groups staggered by one segment (L = DMA issue + fragment reads, C = 32 MFMAs), counted vmcnt waits.
#include <hip/hip_runtime.h>
#include <math.h>

#include "common.h"
#include "engine.h"

namespace dn {

typedef __attribute__((address_space(3))) void* lptr_w;

namespace {

struct WgTnParams {
  const void* dy; int lddy, cout;           // dY [M][lddy] bf16
  const void* x[DN_MAX_TERMS]; int ldx[DN_MAX_TERMS], shift[DN_MAX_TERMS];
  int n_taps, cin, rows_w, n_total;         // n_total = n_taps * rows_w (columns of the partial output)
  int M, T, frames_per_slice;               // frames_per_slice % 32 == 0
  float* out; int64_t out_slice_stride;     // partial sums [slices][cout][n_total] (accumulate == 0)
  float* grad; int Np, Kp;                  // accumulate == 1 (one slice): grad[tap][Np][Kp] += result
  int accumulate;
  // groups (blockIdx.z; the blocks of a WaveNet stack in one launch): element strides between the groups' operands and results;
  // shift_by_group: the taps' shifts are scaled by 2^group (the stack's dilations)
  int64_t dy_gstride, x_gstride[DN_MAX_TERMS], out_gstride, grad_gstride;
  int shift_by_group;
};

__device__ uint4 g_zero_page_w[2];  // 32 bytes of zeros (static storage is zero-initialised)

__device__ __forceinline__ void glds16w(const void* src, uint32_t lds_addr) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_addr))
               : "memory");
}
template <int N>
__device__ __forceinline__ void pipe_sync_w() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

constexpr int TROW = 512;            // bytes of one frame's 256 columns
constexpr int TTILE = 32 * TROW;     // one operand's K-tile: 16 KiB
constexpr int TSTAGE = 2 * TTILE;    // X tile, then dY tile
// ring depth: 4 stages (128 KiB) or 5 (160 KiB: all of a CU's LDS -- the epilogue stores straight from the accumulators).  The K loop
// moves 34 GB/s per CU on the VAE's FFN-conv gradient with 96 KiB in flight; a fifth stage was the test of whether that is latency: it is not.

// byte offset of 8-byte piece `byte` of frame row `row` (0..31) in a tile: 32-byte granules swizzled with the row
__device__ __forceinline__ int tn_off(int row, int byte) { return row * TROW + ((((byte >> 5) ^ (row & 15))) << 5) + (byte & 31); }

// KW: k columns of a tile.  256 = 8 waves of 64 x 128; 192 = 8 waves of 48 x 128 (three A fragments instead of four: 22 transposing
// reads per 24 MFMAs instead of 24 per 32) for the shapes whose 256-wide tiles leave a quarter of the chip idle -- the VAE's FFN conv:
// 6144 x 2048 is 24 x 8 = 192 tiles of 256 x 256 but 32 x 8 = 256 of 192 x 256.  The LDS layout keeps its 512-byte row pitch; the
// X tile's columns beyond 192 are not fetched (zero page).
template <int TSTAGES, int KW = 256>
__global__ __launch_bounds__(512, 1) void wgrad_tn_kernel(const WgTnParams p) {
  constexpr int WKC = KW / 4;   // k columns of a wave: 64 or 48
  constexpr int NA = WKC / 16;  // its A fragments
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave >> 1, wn = wave & 1;  // wave tile: k columns [WKC wk, +WKC) x n columns [128 wn, +128)
  const int k_tiles = (p.n_total + KW - 1) / KW;
  const int kq0 = (blockIdx.x % k_tiles) * KW, n0 = (blockIdx.x / k_tiles) * 256;
  const int slice = blockIdx.y;
  const int g = blockIdx.z;
  const int f_begin = slice * p.frames_per_slice;
  int f_end = f_begin + p.frames_per_slice;
  f_end = f_end < p.M ? f_end : p.M;
  const int nkt = (f_end - f_begin + 31) / 32;
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lptr_w)smem);

  // ---- staging: wave w stages pieces 2w, 2w+1 (two frame rows each) of both tiles.  LDS position (row, 16-byte slot q) <- source
  // 16-byte chunk (((q >> 1) ^ (row & 15)) << 1) | (q & 1) of that row.
  const char* zero_src = reinterpret_cast<const char*>(g_zero_page_w);
  const char* x_ptr[2]; const char* y_ptr[2];
  int x_t[2], x_m[2], x_shift[2];
  int64_t x_inc[2], y_inc = (int64_t)32 * p.lddy * 2;
  bool x_col_ok[2], y_col_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 2 + (lane >> 5), q = lane & 31;
    const int cs = (((q >> 1) ^ (row & 15)) << 1) | (q & 1);
    const int m = f_begin + row;
    // X: column kq0 + 8 cs belongs to tap kq / rows_w
    const int kq = kq0 + cs * 8;
    int tap = kq / p.rows_w;
    const int k = kq - tap * p.rows_w;
    const bool ok = kq < p.n_total && k < p.cin && cs * 8 < KW;  // (a chunk that straddles cin relies on the buffer's zero pad columns)
    tap = tap < p.n_taps ? tap : 0;
    x_shift[i] = p.shift_by_group ? p.shift[tap] << g : p.shift[tap];
    x_col_ok[i] = ok;
    x_m[i] = m;
    x_t[i] = m % p.T;
    x_inc[i] = (int64_t)32 * p.ldx[tap] * 2;
    x_ptr[i] = reinterpret_cast<const char*>(p.x[tap]) + (g * p.x_gstride[tap] + (int64_t)(m - x_shift[i]) * p.ldx[tap] + k) * 2;
    const int n = n0 + cs * 8;
    y_col_ok[i] = n < p.cout;  // (as above for a chunk that straddles cout)
    y_ptr[i] = reinterpret_cast<const char*>(p.dy) + (g * p.dy_gstride + (int64_t)m * p.lddy + n) * 2;
  }
  const int t_step = 32 % p.T;
  auto stage = [&](int slot) {
    const uint32_t xb = lds_base + slot * TSTAGE + wave * 2048, yb = xb + TTILE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool in = x_m[i] < f_end;
      glds16w(in && x_col_ok[i] && x_t[i] >= x_shift[i] ? x_ptr[i] : zero_src, xb + i * 1024);
      glds16w(in && y_col_ok[i] ? y_ptr[i] : zero_src, yb + i * 1024);
      x_ptr[i] += x_inc[i]; y_ptr[i] += y_inc;
      x_m[i] += 32;
      x_t[i] += t_step;
      x_t[i] = x_t[i] >= p.T ? x_t[i] - p.T : x_t[i];
    }
  };

  f32x4 acc[NA][8];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment reads: lane i of a 16-lane group (fg = lane >> 4) supplies frame row fg * 4 + (i >> 2), 8-byte piece (i & 3) of the
  // 32-byte block of its 16 columns; + 16 rows for the upper half of the k-values
  const int fr = lane & 15, fg = lane >> 4;
  const int rrow = fg * 4 + (fr >> 2);
  int a_rd[NA], b_rd[8];
#pragma unroll
  for (int a = 0; a < NA; ++a) a_rd[a] = tn_off(rrow, (wk * WKC + a * 16) * 2 + (fr & 3) * 8);
#pragma unroll
  for (int b = 0; b < 8; ++b) b_rd[b] = TTILE + tn_off(rrow, (wn * 128 + b * 16) * 2 + (fr & 3) * 8);
  uint4 af[NA], bf[8];
  auto tr = [&](const char* base, int off) -> uint2 {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off));
    return __builtin_bit_cast(uint2, v);
  };
  auto load_frags = [&](int slot) {
    const char* sb = smem + slot * TSTAGE;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const uint2 lo = tr(sb, a_rd[a]), hi = tr(sb, a_rd[a] + 16 * TROW);  // (row + 16 keeps row & 15: same swizzle)
      af[a] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const uint2 lo = tr(sb, b_rd[b]), hi = tr(sb, b_rd[b] + 16 * TROW);
      bf[b] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
  };
  auto mma_all = [&]() {
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int a = 0; a < NA; ++a) mma_kstep<BF16>(acc[a][b], af[a], bf[b]);
  };

  constexpr int PER = 4;  // DMA pieces per wave per stage
  const bool late = wave >= 4;
#pragma unroll
  for (int st = 0; st < TSTAGES - 1; ++st)
    if (st < nkt) stage(st);
  // tile 0 has landed; up to TSTAGES - 2 later ones may stay in flight
  if (nkt >= TSTAGES) pipe_sync_w<(TSTAGES - 2) * PER>();
  else if (nkt == 4) pipe_sync_w<3 * PER>();  // (TSTAGES == 5 only)
  else if (nkt == 3) pipe_sync_w<2 * PER>();
  else if (nkt == 2) pipe_sync_w<PER>();
  else pipe_sync_w<0>();
  __builtin_amdgcn_sched_barrier(0);
  if (late) pipe_sync_w<63>();  // the stagger
  auto ktile = [&](auto late_c, auto stage_c, auto sync_c, int slot, int fill) {
    constexpr bool LATE = decltype(late_c)::value, STAGE = decltype(stage_c)::value;
    constexpr int SYNC = decltype(sync_c)::value;
    load_frags(slot);
    if constexpr (STAGE) stage(fill);
    if constexpr (LATE) pipe_sync_w<SYNC>(); else pipe_sync_w<63>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mma_all();
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LATE) pipe_sync_w<63>(); else pipe_sync_w<SYNC>();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto run = [&](auto late_c) {
    using std::integral_constant;
    int slot = 0, fill = TSTAGES - 1, kt = 0;
    auto adv = [&]() { slot = slot == TSTAGES - 1 ? 0 : slot + 1; fill = fill == TSTAGES - 1 ? 0 : fill + 1; };
    // the barrier that closes K-tile kt needs tile kt + 1 landed: tiles kt + 2 .. (last issued) may stay in flight
    for (; kt + (TSTAGES - 1) < nkt; ++kt) { ktile(late_c, std::true_type{}, integral_constant<int, (TSTAGES - 2) * PER>{}, slot, fill); adv(); }
    if constexpr (TSTAGES == 5) {
      if (nkt >= 4) { ktile(late_c, std::false_type{}, integral_constant<int, 2 * PER>{}, slot, fill); adv(); }
    }
    if (nkt >= 3) { ktile(late_c, std::false_type{}, integral_constant<int, PER>{}, slot, fill); adv(); }
    if (nkt >= 2) { ktile(late_c, std::false_type{}, integral_constant<int, 0>{}, slot, fill); adv(); }
    ktile(late_c, std::false_type{}, integral_constant<int, 0>{}, slot, fill);
  };
  if (nkt > 0) {
    if (late) run(std::true_type{}); else run(std::false_type{});
  }
  if (!late) pipe_sync_w<63>();

  // ---- epilogue: acc[a][b][r] = result for k column kq0 + 64 wk + 16 a + 4 fg + r, n = n0 + 128 wn + 16 b + fr
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const int n = n0 + wn * 128 + b * 16 + fr;
    if (n >= p.cout) continue;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const int kq = kq0 + wk * WKC + a * 16 + fg * 4;
      if (kq >= p.n_total) continue;
      const int tap = kq / p.rows_w, k = kq - tap * p.rows_w;
      f32x4 v = acc[a][b];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = k + r < p.cin ? v[r] : 0.f;  // (pad columns of X need not be zero: their products stay in these columns)
      if (p.accumulate) {
        if (k >= p.Kp) continue;
        float4* dst = reinterpret_cast<float4*>(p.grad + g * p.grad_gstride + ((int64_t)tap * p.Np + n) * p.Kp + k);
        float4 g = *dst;
        g.x += v[0]; g.y += v[1]; g.z += v[2]; g.w += v[3];
        *dst = g;
      } else {
        *reinterpret_cast<float4*>(p.out + g * p.out_gstride + slice * p.out_slice_stride + (int64_t)n * p.n_total + kq) =
            make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
}


}  // namespace

// see include/diffnorm_hip.h
int wgrad_tn_launch(const void* dy, int lddy, int cout, const void* const* x, const int* ldx, const int* shift, int n_taps, int cin, int B, int T,
                    int slices, float* part, float* grad, void* stream, int tag, const WgTnGroups* grp) {
  WgTnParams p;
  memset(&p, 0, sizeof(p));
  DN_CHECK_ARG(dy && x && ldx && shift && n_taps >= 1 && n_taps <= DN_MAX_TERMS && cin > 0 && cout > 0 && B > 0 && T > 0 && slices >= 1,
               "dn_conv_weight_grad_tn: bad arguments");
  DN_CHECK_ARG(lddy % 8 == 0 && lddy >= cout, "dn_conv_weight_grad_tn: dY rows must be 16-byte multiples covering cout (lddy=%d)", lddy);
  p.dy = dy; p.lddy = lddy; p.cout = cout;
  for (int j = 0; j < n_taps; ++j) {
    DN_CHECK_ARG(x[j] && ldx[j] % 8 == 0 && ldx[j] >= cin && shift[j] >= 0, "dn_conv_weight_grad_tn: tap %d: ldx=%d shift=%d", j, ldx[j], shift[j]);
    p.x[j] = x[j]; p.ldx[j] = ldx[j]; p.shift[j] = shift[j];
  }
  p.n_taps = n_taps; p.cin = cin; p.rows_w = padn(cin); p.n_total = n_taps * p.rows_w;
  p.M = B * T; p.T = T;
  p.frames_per_slice = ((p.M + slices - 1) / slices + 31) / 32 * 32;
  DN_CHECK_ARG((slices == 1) == (part == nullptr) || grad == nullptr, "dn_conv_weight_grad_tn: one slice accumulates into grad, several write part");
  p.accumulate = part == nullptr;
  DN_CHECK_ARG(p.accumulate ? (grad != nullptr && slices == 1) : true, "dn_conv_weight_grad_tn: accumulation needs grad and one slice");
  p.out = part; p.out_slice_stride = (int64_t)cout * p.n_total;
  p.grad = grad; p.Np = padn(cout); p.Kp = padk(cin);
  int groups = 1;
  if (grp && grp->groups > 1) {  // part: [group][slice][cout][n_total]
    groups = grp->groups;
    p.dy_gstride = grp->dy_gstride;
    for (int j = 0; j < n_taps; ++j) p.x_gstride[j] = grp->x_gstride;
    p.out_gstride = (int64_t)slices * p.out_slice_stride;
    p.grad_gstride = grp->grad_gstride;
    p.shift_by_group = grp->shift_by_group;
  }
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_tn_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TSTAGE);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_tn_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * TSTAGE);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_tn_kernel<4, 192>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TSTAGE);
    attr_done = true;
  }
  // k columns of a tile: option wgrad_k192 = 1 -> 192 where that fills the chip better than 256.  Off: measured on the VAE's FFN conv
  // (192 -> 256 workgroups) the kernel alone gains 7 % (352 -> 328 us, 0.35 -> 0.37 of peak), but the update LOSES 1.7 % (16.11 ->
  // 16.39 ms, three alternating pairs): the kernel runs on the second stream beside the data-gradient chain, and a launch that takes
  // every CU for 330 us starves the chain that 192 workgroups left 64 CUs to.
  const long n_tiles = (cout + 255) / 256, per = (long)slices * groups;
  auto fill = [](long tiles) { const double r = (double)tiles / 256.0; return r / ceil(r); };
  const long t256 = ((p.n_total + 255) / 256) * n_tiles * per, t192 = ((p.n_total + 191) / 192) * n_tiles * per;
  // per output a 192-wide tile costs 22 / 24 transposing reads per MFMA against 24 / 32: worth 0.82 of the 256-wide one on a full chip
  const bool k192 = option_or(OPT_WGRAD_K192, 0) != 0 && per == 1 /* (the sliced launches' slice counts were chosen for 256-wide tiles) */ && 0.82 * fill(t192) * p.n_total / (192.0 * ((p.n_total + 191) / 192)) >
                                                             fill(t256) * p.n_total / (256.0 * ((p.n_total + 255) / 256));
  dim3 grid((k192 ? (p.n_total + 191) / 192 : (p.n_total + 255) / 256) * (unsigned)n_tiles, slices, groups);
  const bool timed = tag != 0 && g_prof.cap > 0 && tag == g_prof.tag && g_prof.n < g_prof.cap;
  // option wgrad_stages = 5: the 160 KiB ring (A/B timing: measured level with the 128 KiB one, 353 vs 354 us on the VAE's FFN-conv
  // gradient -- the loop is not waiting for its operands).  Also measured and dropped (round 4): four waves of 128 x 128 with the
  // fragments double-buffered in registers and the accumulators pinned to AGPRs, 16 fragments per 64 MFMAs instead of 12 per 32 --
  // 434 vs 343 us: with one wave per SIMD the compiler's schedule leaves the LDS round trips of a K-tile exposed.
  const bool deep = option_or(OPT_WGRAD_STAGES, 4) >= 5;
  if (timed) (void)hipEventRecord(g_prof.ev[2 * g_prof.n], (hipStream_t)stream);
  if (k192) hipLaunchKernelGGL((wgrad_tn_kernel<4, 192>), grid, dim3(512), 4 * TSTAGE, (hipStream_t)stream, p);
  else if (deep) hipLaunchKernelGGL(wgrad_tn_kernel<5>, grid, dim3(512), 5 * TSTAGE, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(wgrad_tn_kernel<4>, grid, dim3(512), 4 * TSTAGE, (hipStream_t)stream, p);
  if (timed) (void)hipEventRecord(g_prof.ev[2 * g_prof.n++ + 1], (hipStream_t)stream);
  DN_CHECK_LAUNCH("dn_conv_weight_grad_tn");
  return DN_OK;
}

}  // namespace dn

extern "C" int dn_conv_weight_grad_tn(const void* dy, int32_t lddy, int32_t cout, const void* const* x, const int32_t* ldx, const int32_t* shift,
                                      int32_t n_taps, int32_t cin, int32_t B, int32_t T, int32_t slices, float* part, float* grad, void* stream) {
  return dn::wgrad_tn_launch(dy, lddy, cout, x, ldx, shift, n_taps, cin, B, T, slices, part, grad, stream, 0, nullptr);
}
