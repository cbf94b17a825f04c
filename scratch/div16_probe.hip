// Exhaustive probe on the GPU: is the float16 quotient h(v / s) reproduced by packed float16 arithmetic
//   v' = v * P,  q0 = v' * r,  rem = fma(-q0, s', v'),  q1 = fma(rem, r, q0)
// with s' = s * P (P a power of two that brings s' into [1, 2)) and r = RN16(1 / s')?
// Every finite v against every s in [2^-15, 0x70E2].  Reference: float32 division of the float16 values rounded to
// float16 (correctly rounded: 24 >= 2 * 11 + 2 bits).  Reports per variant the number of differing pairs, how many
// of them change rint(q) for |q| <= 600 and the s values involved.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__device__ half_t hb(uint32_t b) { return __builtin_bit_cast(half_t, (uint16_t)b); }
__device__ uint32_t bh(half_t h) { return __builtin_bit_cast(uint16_t, h); }

// counts[s_index][0..2]: differing quotients, differing rint for |q| <= 600, differing with |q| >= 2^-4
__global__ void k_probe(uint32_t s_lo, uint32_t s_hi, unsigned long long* counts, int variant) {
  const uint32_t sb = s_lo + blockIdx.x;
  if (sb > s_hi) return;
  const half_t s = hb(sb);
  const float sf = (float)s;
  // P = 2^k with s * P in [1, 2)
  int e;
  frexpf(sf, &e);           // sf = m * 2^e, m in [0.5, 1)
  const int k = 1 - e;      // s * 2^k in [1, 2)
  const float Pf = ldexpf(1.0f, k);
  const half_t P = (half_t)Pf;         // representable for k in [-14, 15]
  const half_t s1 = (half_t)(sf * Pf); // exact
  const half_t r = (half_t)(1.0f / (float)s1);
  unsigned long long bad = 0, bad_rint = 0, bad_big = 0;
  for (uint32_t vb = threadIdx.x; vb < 0x7c00u; vb += blockDim.x) {
    const half_t v = hb(vb);
    const half_t want = (half_t)((float)v / sf);
    half_t v1 = v * P;
    half_t q0 = v1 * r;
    half_t rem = __builtin_fmaf16(-q0, s1, v1);
    half_t q1 = __builtin_fmaf16(rem, r, q0);
    if (variant == 1) {  // a second correction step
      rem = __builtin_fmaf16(-q1, s1, v1);
      q1 = __builtin_fmaf16(rem, r, q1);
    }
    const float a = fabsf((float)want);
    if (bh(q1) != bh(want) && a <= 600.f) {
      if (a < 0.001953125f) ++bad;            // below 2^-9
      else if (a < 0.0625f) ++bad_rint;       // 2^-9 .. 2^-4
      else ++bad_big;                         // 2^-4 .. 600
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    bad += __shfl_xor(bad, off); bad_rint += __shfl_xor(bad_rint, off); bad_big += __shfl_xor(bad_big, off);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&counts[3 * blockIdx.x + 0], bad);
    atomicAdd(&counts[3 * blockIdx.x + 1], bad_rint);
    atomicAdd(&counts[3 * blockIdx.x + 2], bad_big);
  }
}

int main() {
  const uint32_t s_lo = 0x0200u, s_hi = 0x70E2u;  // 2^-15 .. the upper clamp of the scale
  const int ns = s_hi - s_lo + 1;
  unsigned long long* d;
  hipMalloc(&d, ns * 3 * 8);
  unsigned long long* h = (unsigned long long*)malloc(ns * 3 * 8);
  for (int variant = 0; variant < 2; ++variant) {
    hipMemset(d, 0, ns * 3 * 8);
    hipLaunchKernelGGL(k_probe, dim3(ns), dim3(256), 0, 0, s_lo, s_hi, d, variant);
    hipMemcpy(h, d, ns * 3 * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    int s_bad = 0, s_bad_big = 0, shown = 0;
    for (int i = 0; i < ns; ++i) {
      t0 += h[3 * i]; t1 += h[3 * i + 1]; t2 += h[3 * i + 2];
      if (h[3 * i]) ++s_bad;
      if (h[3 * i + 2] || h[3 * i + 1]) {
        ++s_bad_big;
        if (shown < 12) { printf("  s = 0x%04x: below 2^-9 %llu, 2^-9..2^-4 %llu, 2^-4..600 %llu\n", s_lo + i, h[3 * i], h[3 * i + 1], h[3 * i + 2]); ++shown; }
      }
    }
    printf("variant %d: differing pairs with |q| < 2^-9: %llu, 2^-9..2^-4: %llu, 2^-4..600: %llu (of %llu); scales with a difference at |q| >= 2^-9: %d of %d\n",
           variant, t0, t1, t2, (unsigned long long)ns * 0x7c00ull, s_bad_big, ns);
  }
  return 0;
}
