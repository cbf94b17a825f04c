// Memory floor of the c2 traffic mix: per 8 elements read 16 B, write 16 B (deq) + 8 B (codes) + 32 B (|err| f32),
// no arithmetic to speak of.  Variants: grid size, loads in flight, |err| store pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int NV, bool CONTIG>
__global__ __launch_bounds__(256) void k_stream(const u32x4* __restrict__ w, u32x4* __restrict__ deq, u32x2* __restrict__ codes,
                                                u32x4* __restrict__ err, int64_t nvec) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  const int64_t tiles = nvec / (64 * NV);
  u32x4 nxt[NV];
  if (gw < tiles)
    for (int j = 0; j < NV; ++j) nxt[j] = w[(gw * NV + j) * 64 + lane];
  for (int64_t t = gw; t < tiles; t += nw) {
    u32x4 raw[NV];
    for (int j = 0; j < NV; ++j) raw[j] = nxt[j];
    if (t + nw < tiles)
      for (int j = 0; j < NV; ++j) nxt[j] = w[((t + nw) * NV + j) * 64 + lane];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int64_t v = (t * NV + j) * 64 + lane;
      u32x4 r = raw[j];
      deq[v] = r ^ 0x00010001u;
      codes[v] = u32x2{r[0] + r[1], r[2] + r[3]};
      u32x4 a0 = {r[0] << 16, r[0] & 0xffff0000u, r[1] << 16, r[1] & 0xffff0000u};
      u32x4 a1 = {r[2] << 16, r[2] & 0xffff0000u, r[3] << 16, r[3] & 0xffff0000u};
      if (CONTIG) {
        const int64_t base = (t * NV + j) * 128;
        err[base + lane] = a0;
        err[base + 64 + lane] = a1;
      } else {
        err[2 * v] = a0;
        err[2 * v + 1] = a1;
      }
    }
  }
}

int main() {
  const int64_t n = 16ll * 4096 * 4096, nvec = n / 8;
  void *w, *deq, *codes, *err;
  hipMalloc(&w, n * 2); hipMalloc(&deq, n * 2); hipMalloc(&codes, n); hipMalloc(&err, n * 4);
  hipMemset(w, 1, n * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double bytes = n * 9.0;
  auto run = [&](const char* name, auto kern, int grid) {
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, (const u32x4*)w, (u32x4*)deq, (u32x2*)codes, (u32x4*)err, nvec);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    printf("%-28s grid %5d  %7.1f us  %6.2f TB/s\n", name, grid, best * 1e3, bytes / best / 1e9);
  };
  for (int grid : {1024, 2048, 4096, 16384}) {
    run("NV4 split |err| stores", k_stream<4, false>, grid);
    run("NV4 contiguous |err|", k_stream<4, true>, grid);
    run("NV8 contiguous |err|", k_stream<8, true>, grid);
    run("NV2 contiguous |err|", k_stream<2, true>, grid);
  }
  return 0;
}
