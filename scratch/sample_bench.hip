// Micro-benchmark: where does the time of a sampled LDS histogram go?  (loads vs LDS atomics vs flush)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define BINS 32768
__device__ __forceinline__ uint32_t sample_line(uint32_t c) { return (c * 0x9E3779B1u) >> 27; }

// mode bit0: do loads (else synthesize), bit1: do LDS atomics, bit2: flush to global, bit3: zero LDS
template <int U>
__global__ __launch_bounds__(1024) void k_sample(const u32x4* x, int64_t n, uint32_t* g, int mode, int contiguous) {
  extern __shared__ uint32_t lh[];
  if (mode & 8) for (int i = threadIdx.x; i < BINS; i += blockDim.x) lh[i] = 0;
  __syncthreads();
  const int64_t nchunks = n / 2048, total = nchunks * 16;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (int64_t i0 = t0; i0 < total; i0 += U * step) {
    u32x4 raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * step, c = i >> 4;
      raw[u] = u32x4{0x39000000u + (uint32_t)i * 977u % 0x20000u, 0x39010000u, 0x39020000u + threadIdx.x, 0x39030000u};
      if ((mode & 1) && i < total)
        raw[u] = contiguous ? x[i] : x[c * 512 + sample_line((uint32_t)c) * 16 + ((uint32_t)i & 15u)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i0 + u * step < total) {
        if (mode & 2) {
#pragma unroll
          for (int k = 0; k < 4; ++k) atomicAdd(&lh[(raw[u][k] >> 16) & 0x7fffu], 1u);
        } else {
          acc += raw[u][0] ^ raw[u][1] ^ raw[u][2] ^ raw[u][3];
        }
      }
    }
  }
  __syncthreads();
  if (mode & 4) {
    for (int i = threadIdx.x; i < BINS; i += blockDim.x) {
      uint32_t c = lh[i];
      if (c) { atomicAdd(&g[i], c); atomicAdd(&g[BINS + (i >> 8)], c); }
    }
  }
  if (acc == 0x12345678u) g[0] = acc;
}

int main() {
  const int64_t n = 16ll * 4096 * 4096;  // 16 arrays back to back, treated as one
  std::vector<float> h(n);
  uint64_t s = 88172645463325252ull;
  for (int64_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (float)((s >> 40) * (1.0 / 16777216.0)) * 2.3e-4f; }
  float* x; uint32_t* g;
  hipMalloc(&x, n * 4); hipMalloc(&g, (BINS + 128) * 4);
  hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMemset(g, 0, (BINS + 128) * 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_sample<8>), hipFuncAttributeMaxDynamicSharedMemorySize, BINS * 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_sample<1>), hipFuncAttributeMaxDynamicSharedMemorySize, BINS * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, int U, int mode, int contiguous, int grid) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      if (U == 8) hipLaunchKernelGGL(k_sample<8>, dim3(grid), dim3(1024), BINS * 4, 0, (const u32x4*)x, n, g, mode, contiguous);
      else hipLaunchKernelGGL(k_sample<1>, dim3(grid), dim3(1024), BINS * 4, 0, (const u32x4*)x, n, g, mode, contiguous);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    printf("%-44s U=%d grid=%d  %8.1f us\n", name, U, grid, best * 1e3);
  };
  for (int grid : {256, 512}) {
    run("empty (zero LDS only)", 8, 8, 0, grid);
    run("zero + flush", 8, 8 | 4, 0, grid);
    run("loads only, scattered", 8, 1, 0, grid);
    run("loads only, scattered", 1, 1, 0, grid);
    run("loads only, contiguous 1/32", 8, 1, 1, grid);
    run("atomics only (synthetic, spread bins)", 8, 8 | 2, 0, grid);
    run("loads + atomics scattered", 8, 8 | 1 | 2, 0, grid);
    run("loads + atomics contiguous", 8, 8 | 1 | 2, 1, grid);
    run("all scattered", 8, 15, 0, grid);
    run("all scattered", 1, 15, 0, grid);
  }
  return 0;
}
