// What does one LDS-DMA instruction (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB) cost the wave that issues it, and does
// the cost depend on how the 64 lane addresses fall on cache lines?  The hand-scheduled contraction kernel pays ~50 cycles per
// piece beyond its MFMA gap; its pieces are 16 rows x 64 B (16 half-lines per instruction).  This times, with the
// cycle counter, batches of 8 back-to-back DMA instructions for pieces of 16 x 64 B, 8 x 128 B, 4 x 256 B and 1 x 1024 B, with
// 1..4 waves of the workgroup (one per SIMD) issuing at the same time.
//   hipcc -O3 --offload-arch=gfx950 tools/dma_issue_cost.hip -o tools/dma_issue_cost && tools/dma_issue_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
// the same 16 bytes per lane into VGPRs instead of LDS (register staging: the data would then go to LDS with ds_write_b128)
__device__ __forceinline__ void vload16(const void* src, u32x4_t& dst) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
}

__device__ __forceinline__ void dma16(const void* src, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_addr) : "memory");
}

// rows_log2: lanes per row = 64 >> rows_log2 ... a piece is (1 << rows_log2) rows of (1024 >> rows_log2) bytes
__global__ __launch_bounds__(256) void dma_cost(const char* buf, int64_t row_stride, int rows_log2, int active_waves, int iters,
                                                int pieces_per_region, int mode, uint64_t* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lanes_per_row = 64 >> rows_log2;
  int row = lane / lanes_per_row, chunk = lane % lanes_per_row;
  if (mode == 2) { row = (lane & 31) >> 2; chunk = (lane & 3) + 4 * (lane >> 5); }  // 8 rows x 128 B as two half-row lane groups
  const int rows = 1 << rows_log2;
  // each wave of each workgroup walks its own region of pieces_per_region pieces, over and over (L2-resident)
  const int64_t region = (int64_t)(blockIdx.x * 4 + wave) * pieces_per_region * rows * row_stride;
  const char* p0 = buf + region + row * row_stride + chunk * 16;
  const int64_t piece_stride = rows * row_stride;
  const uint32_t lds = (uint32_t)(uintptr_t)smem + wave * 8192;
  uint64_t issue = 0, total = 0;
  uint32_t sink = 0;
  const bool vsink = iters < 0;  // never: keeps the loaded registers alive without costing the loop anything
  if (wave < active_waves) {
    int pc = 0;
    for (int it = 0; it < iters; ++it) {
      // mode 1: 16 x 64 B pieces that read the two halves of each 128 B line in consecutive batches (consecutive K-tiles)
      const char* p = p0 + pc * piece_stride + (mode == 1 ? (it & 1) * 64 : 0);
      const uint64_t c0 = __builtin_readcyclecounter();
      uint64_t c1;
      if (mode == 3) {
        u32x4_t r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) vload16(p + i * piece_stride, r[i]);
        c1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : : "memory");
        if (vsink) {
#pragma unroll
          for (int i = 0; i < 8; ++i) sink ^= r[i][0];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) dma16(p + i * piece_stride, lds + i * 1024);
        c1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      const uint64_t c2 = __builtin_readcyclecounter();
      if (it >= 4) { issue += c1 - c0; total += c2 - c0; }
      if (mode != 1 || (it & 1)) pc += 8;
      if (pc + 8 > pieces_per_region) pc = 0;
    }
  }
  if (sink == 0x12345678u) out[0] = sink;
  if (lane == 0) {
    out[(blockIdx.x * 4 + wave) * 2] = issue;
    out[(blockIdx.x * 4 + wave) * 2 + 1] = total;
  }
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const int64_t row_stride = 2816;  // bytes between rows: K = 1408 bf16, the FFN conv's operands
  const int pieces = 32, iters = 204;
  const size_t bytes = (size_t)cus * 4 * pieces * 16 * row_stride + 4096;
  char* buf; uint64_t* out;
  hipMalloc(&buf, bytes);
  hipMemset(buf, 1, bytes);
  hipMalloc(&out, sizeof(uint64_t) * cus * 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(dma_cost), hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
  const char* names[] = {"1 x 1024 B", "2 x 512 B", "4 x 256 B", "8 x 128 B", "16 x 64 B", "16x64 halves", "8x128 split",
                         "VGPR 8x128 B", "VGPR 16x64 B"};
  for (int active = 1; active <= 4; active += 3)
    for (int cfg = 0; cfg <= 8; ++cfg) {
      const int rl = cfg <= 4 ? cfg : cfg == 5 ? 4 : cfg == 8 ? 4 : 3, mode = cfg <= 4 ? 0 : cfg >= 7 ? 3 : cfg - 4;
      for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(dma_cost, dim3(cus), dim3(256), 32768, 0, buf, row_stride, rl, active, iters, pieces, mode, out);
        hipDeviceSynchronize();
      }
      std::vector<uint64_t> h(cus * 8);
      hipMemcpy(h.data(), out, sizeof(uint64_t) * cus * 8, hipMemcpyDeviceToHost);
      double is = 0, tot = 0;
      int n = 0;
      for (int b = 0; b < cus; ++b)
        for (int w = 0; w < active; ++w) { is += h[(b * 4 + w) * 2]; tot += h[(b * 4 + w) * 2 + 1]; ++n; }
      const double per = (double)(iters - 4) * 8 * n;
      printf("piece %-11s  %d wave(s)/CU issuing: %6.1f cycles to issue one instruction, %7.1f per instruction incl. completion "
             "(8 in flight)\n", names[cfg], active, is / per, tot / per);
    }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { printf("HIP error: %s\n", hipGetErrorString(e)); return 1; }
  return 0;
}
