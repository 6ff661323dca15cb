// Which ds_read_b128 address patterns of an MFMA B-fragment read (lane l: row l % 16 + shift, 16-byte chunk l / 16 of a 64-byte row,
// chunk position swizzled by a function of the row) are free of LDS bank conflicts?  Prints cycles per instruction for row shifts
// 0..3 and a set of candidate swizzles.  Build: hipcc --offload-arch=gfx950 -O3 -o lds_conflict_probe lds_conflict_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ int swz(int id, int fq, int row) {
  switch (id) {
    case 0: return fq ^ (((row >> 3) & 1) << 1);           // the kernels' swizzle
    case 1: return fq ^ ((row >> 2) & 3);
    case 2: return fq ^ ((row >> 1) & 3);
    case 3: return (fq + (row >> 2)) & 3;
    case 4: return fq ^ (row & 3);
    case 5: return fq;                                      // none
    case 6: return fq ^ ((row >> 3) & 3);
    case 7: return fq ^ (((row >> 2) & 1) << 1) ^ ((row >> 3) & 1);
    case 8: return fq ^ (((row >> 3) & 1) << 1) ^ ((row >> 2) & 1);
    default: return fq ^ (((row >> 1) & 1) << 1) ^ ((row >> 2) & 1);
  }
}

__global__ void probe(int id, int shift, int iters, uint64_t* out, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = i;
  __syncthreads();
  const int frow = lane & 15, fq = lane >> 4;
  const int row = 16 + frow - shift;
  const uint32_t addr = row * 64 + swz(id, fq, row) * 16;
  uint4 acc = make_uint4(0, 0, 0, 0);
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint4 v;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(j * 2048));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  sink[threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

// throughput form: many reads in flight (the latency-bound loop above hides 2-way conflicts behind the round trip)
__global__ void probe_tp(int id, int shift, int iters, uint64_t* out, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = i;
  __syncthreads();
  const int frow = lane & 15, fq = lane >> 4;
  const int row = 16 + frow - shift;
  const uint32_t addr = row * 64 + swz(id, fq, row) * 16;
  uint4 acc = make_uint4(0, 0, 0, 0);
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    uint4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[j]) : "v"(addr), "n"(j * 2048));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w; }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  sink[threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

int main() {
  uint64_t* out; uint32_t* sink;
  hipMalloc(&out, 8); hipMalloc(&sink, 4096);
  const int iters = 2000;
  for (int waves = 1; waves <= 4; waves *= 4) {
    printf("waves per workgroup: %d  (cycles per ds_read_b128 per wave, throughput form)\n       ", waves);
    for (int s = 0; s < 4; ++s) printf(" shift%d", s);
    printf("\n");
    for (int id = 0; id < 10; ++id) {
      printf("swz %d: ", id);
      for (int s = 0; s < 4; ++s) {
        hipLaunchKernelGGL(probe_tp, dim3(1), dim3(64 * waves), 32768, 0, id, s, iters, out, sink);
        hipDeviceSynchronize();
        uint64_t c; hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
        printf(" %6.1f", (double)c / (iters * 8));
      }
      printf("\n");
    }
  }
  return 0;
}
