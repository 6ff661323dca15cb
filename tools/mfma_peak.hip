// Measures what the matrix pipe of this particular GPU sustains: a register-only loop of independent bf16 MFMAs
// (no LDS, no memory) on every SIMD, timed with HIP events, plus the shader clock it ran at (s_memtime ticks per
// s_memrealtime tick).  Gives the practical ceiling the contraction kernels are compared with in DESIGN.md.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak && tools/mfma_peak [waves_per_simd] [ms]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(int iters, uint64_t* clk, float* sink) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(0.5f - i * 0.01f); }
  const uint64_t c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  const uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) sink[0] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 2;       // waves per SIMD
  const double target_ms = argc > 2 ? atof(argv[2]) : 20;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const int threads = 64 * 4 * wps;
  constexpr int NACC = 16;
  uint64_t* clk; float* sink;
  hipMalloc(&clk, sizeof(uint64_t) * 2 * cus);
  hipMalloc(&sink, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 2000;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(cus), dim3(threads), 0, 0, iters, clk, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(2 * cus);
    hipMemcpy(h.data(), clk, sizeof(uint64_t) * 2 * cus, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < cus; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
    const double flops = 2.0 * 16 * 16 * 32 * NACC * (double)iters * (threads / 64) * cus;
    printf("waves/SIMD %d iters %d: %.3f ms  %.1f TFLOP/s  shader clock %.0f MHz (cycle counter / 100 MHz real-time counter)  %.2f cycles per MFMA per SIMD\n",
           wps, iters, ms, flops / ms * 1e-9, cyc / rt * 100.0, (cyc / cus) / ((double)iters * NACC * wps));
    if (ms < target_ms) iters = (int)(iters * target_ms / (ms > 0.01 ? ms : 0.01)) + 1;
  }
  return 0;
}
