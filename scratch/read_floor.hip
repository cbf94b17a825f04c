// Read floor of the statistics-only passes: 16 bytes per lane per load, nothing stored.  Variants: grid size,
// vectors in flight per lane, plain vs non-temporal loads, contiguous 1 KiB per wave instruction vs the
// "four vectors of one 64-byte run per lane" pattern of k_quant_team4<T = 1>, and a dummy VALU load of
// WORK packed instructions per 8-element vector to see how much arithmetic hides under the memory time.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// the same read with a small store per 4 KiB step: SB bytes per lane from the first LANES lanes (8 x 64: the
// float64 per-group sums of g = 32; 8 x 16: g = 128; 4 x 64: float32 sums of g = 32)
template <int SB, int LANES, bool NTS>
__global__ __launch_bounds__(256) void k_read_store(const u32x4* __restrict__ w, char* __restrict__ st, int64_t nvec) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  const int64_t tiles = nvec / 256;
  u32x4 nxt[4];
  if (gw < tiles)
    for (int j = 0; j < 4; ++j) nxt[j] = __builtin_nontemporal_load(&w[(gw * 4 + j) * 64 + lane]);
  for (int64_t t = gw; t < tiles; t += nw) {
    u32x4 raw[4];
    for (int j = 0; j < 4; ++j) raw[j] = nxt[j];
    if (t + nw < tiles)
      for (int j = 0; j < 4; ++j) nxt[j] = __builtin_nontemporal_load(&w[((t + nw) * 4 + j) * 64 + lane]);
    uint32_t a = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) a ^= raw[j][0] ^ raw[j][1] ^ raw[j][2] ^ raw[j][3];
    if (lane < LANES) {
      char* dst = st + (t * LANES + lane) * SB;
      if constexpr (SB == 8) {
        uint64_t v = ((uint64_t)a << 32) | a;
        if (NTS) __builtin_nontemporal_store(v, (uint64_t*)dst); else *(uint64_t*)dst = v;
      } else {
        if (NTS) __builtin_nontemporal_store(a, (uint32_t*)dst); else *(uint32_t*)dst = a;
      }
    }
  }
}


// the 4 B x 64 lanes per 4 KiB results kept in registers and written at the end: a wave takes K CONSECUTIVE
// 4 KiB steps, so that its K x 256 B of results are one contiguous run
template <int K, bool NT>
__global__ __launch_bounds__(256) void k_read_store_deferred(const u32x4* __restrict__ w, char* __restrict__ st, int64_t nvec) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  const int64_t tiles = nvec / 256, chunks = tiles / K;
  for (int64_t c = gw; c < chunks; c += nw) {
    uint32_t res[K];
    u32x4 nxt[4];
    for (int j = 0; j < 4; ++j) nxt[j] = NT ? __builtin_nontemporal_load(&w[((c * K) * 4 + j) * 64 + lane]) : w[((c * K) * 4 + j) * 64 + lane];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      u32x4 raw[4];
      for (int j = 0; j < 4; ++j) raw[j] = nxt[j];
      if (k + 1 < K)
        for (int j = 0; j < 4; ++j) nxt[j] = NT ? __builtin_nontemporal_load(&w[((c * K + k + 1) * 4 + j) * 64 + lane]) : w[((c * K + k + 1) * 4 + j) * 64 + lane];
      uint32_t a = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) a ^= raw[j][0] ^ raw[j][1] ^ raw[j][2] ^ raw[j][3];
      res[k] = a;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) *(uint32_t*)(st + ((c * K + k) * 64 + lane) * 4) = res[k];
  }
}

// MAP 0: lane l reads vector j * 64 + l (1 KiB per wave instruction); 1: lane l reads the 64-byte run l;
// 2: lane a + 16 i reads vector j * 64 + 4 a + i (the same 1 KiB per instruction, lanes permuted so that the four
// vectors of a 64-byte run sit 16 lanes apart: a 4 x 4 transpose by v_permlane16/32_swap then gives each lane a run)
template <int NV, int WORK, bool NT, int MAP>
__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ w, uint32_t* __restrict__ out, int64_t nvec) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  const int64_t tiles = nvec / (64 * NV);
  auto ld = [&](int64_t t, int j) {
    const int64_t v = MAP == 1 ? (t * 64 + lane) * NV + j : MAP == 2 ? (t * NV + j) * 64 + 4 * (lane & 15) + (lane >> 4) : (t * NV + j) * 64 + lane;
    return NT ? __builtin_nontemporal_load(&w[v]) : w[v];
  };
  u32x4 nxt[NV];
  if (gw < tiles)
    for (int j = 0; j < NV; ++j) nxt[j] = ld(gw, j);
  uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (int64_t t = gw; t < tiles; t += nw) {
    u32x4 raw[NV];
    for (int j = 0; j < NV; ++j) raw[j] = nxt[j];
    if (t + nw < tiles)
      for (int j = 0; j < NV; ++j) nxt[j] = ld(t + nw, j);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      u32x4 r = raw[j];
      a0 ^= r[0]; a1 ^= r[1]; a2 ^= r[2]; a3 ^= r[3];
#pragma unroll
      for (int k = 0; k < WORK / 4; ++k) {
        asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %5\n v_pk_add_u16 %2, %2, %6\n v_pk_add_u16 %3, %3, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]));
      }
    }
  }
  uint32_t a = a0 ^ a1 ^ a2 ^ a3;
  if (a == 0x12345678u) out[threadIdx.x] = a;
}

int main() {
  const int64_t n = 16ll * 4096 * 4096, nvec = n / 8;
  void *w, *out;
  hipMalloc(&w, n * 2); hipMalloc(&out, 4096);
  hipMemset(w, 1, n * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double bytes = n * 2.0;
  auto run = [&](const char* name, auto kern, int grid) {
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, (const u32x4*)w, (uint32_t*)out, nvec);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    printf("%-34s grid %5d  %7.1f us  %6.2f TB/s\n", name, grid, best * 1e3, bytes / best / 1e9);
  };
  {
    void* st; hipMalloc(&st, n / 4);
    auto run2 = [&](const char* name, auto kern, int grid) {
      float best = 1e9;
      for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, (const u32x4*)w, (char*)st, nvec);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
      }
      printf("%-44s grid %5d  %7.1f us  %6.2f TB/s of reads\n", name, grid, best * 1e3, bytes / best / 1e9);
    };
    for (int grid : {4096, 8192}) {
      run2("read + 8 B x 64 lanes per 4 KiB", k_read_store<8, 64, false>, grid);
      run2("read + 8 B x 64 lanes per 4 KiB, nt store", k_read_store<8, 64, true>, grid);
      run2("read + 4 B x 64 lanes per 4 KiB", k_read_store<4, 64, false>, grid);
      run2("read + 8 B x 16 lanes per 4 KiB", k_read_store<8, 16, false>, grid);
      run2("read + 4 B x 16 lanes per 4 KiB", k_read_store<4, 16, false>, grid);
      run2("read + 8 B x 1 lane per 4 KiB", k_read_store<8, 1, false>, grid);
      run2("deferred 4 B x 64, K = 4 consecutive steps", k_read_store_deferred<4, true>, grid);
      run2("deferred 4 B x 64, K = 8", k_read_store_deferred<8, true>, grid);
      run2("deferred 4 B x 64, K = 16", k_read_store_deferred<16, true>, grid);
      run2("deferred 4 B x 64, K = 16, plain loads", k_read_store_deferred<16, false>, grid);
    }
  }
  for (int grid : {8192}) {
    run("NV4 contiguous", k_read<4, 0, false, false>, grid);
    run("NV4 contiguous nt", k_read<4, 0, true, false>, grid);
    run("NV8 contiguous", k_read<8, 0, false, false>, grid);
    run("NV2 contiguous", k_read<2, 0, false, false>, grid);
    run("NV4 64-byte runs per lane", k_read<4, 0, false, true>, grid);
    run("NV4 permuted lanes (16 apart)", k_read<4, 0, false, 2>, grid);
    run("NV4 permuted lanes (16 apart) nt", k_read<4, 0, true, 2>, grid);
    run("NV4 permuted nt + 64 VALU / vector", k_read<4, 64, true, 2>, grid);
    run("NV4 contiguous nt + 64 VALU / vector", k_read<4, 64, true, 0>, grid);
    run("NV4 runs + 64 VALU / vector", k_read<4, 64, false, 1>, grid);
  }
  for (int grid : {4096, 8192}) {
    run("NV4 + 16 VALU / vector", k_read<4, 16, false, false>, grid);
    run("NV4 + 32 VALU / vector", k_read<4, 32, false, false>, grid);
    run("NV4 + 48 VALU / vector", k_read<4, 48, false, false>, grid);
    run("NV4 + 64 VALU / vector", k_read<4, 64, false, false>, grid);
    run("NV4 + 80 VALU / vector", k_read<4, 80, false, false>, grid);
    run("NV4 + 96 VALU / vector", k_read<4, 96, false, false>, grid);
    run("NV4 runs + 64 VALU / vector", k_read<4, 64, false, true>, grid);
  }
  return 0;
}
