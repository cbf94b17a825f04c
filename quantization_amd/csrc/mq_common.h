// Shared host/device helpers of libmq (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mq.h"

typedef _Float16 half_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

#define MQ_WAVE 64
#define MQ_MAX_GRID 16384  // upper bound of any grid.x (the per-workgroup partial sums live in the workspace)

// ---- host side error plumbing -------------------------------------------------------------------
void mq_set_hip_error(hipError_t e, const char* what);
#define MQ_HIP(call)                          \
  do {                                        \
    hipError_t e_ = (call);                   \
    if (e_ != hipSuccess) {                   \
      mq_set_hip_error(e_, #call);            \
      return MQ_ERR_HIP;                      \
    }                                         \
  } while (0)

// ---- device helpers -----------------------------------------------------------------------------
__device__ __forceinline__ half_t h_from_bits(uint32_t b) { return __builtin_bit_cast(half_t, (uint16_t)b); }
__device__ __forceinline__ uint32_t h_bits(half_t h) { return (uint32_t)__builtin_bit_cast(uint16_t, h); }
__device__ __forceinline__ float f_from_bits(uint32_t b) { return __builtin_bit_cast(float, b); }
__device__ __forceinline__ uint32_t f_bits(float f) { return __builtin_bit_cast(uint32_t, f); }
// float32 -> float16, round to nearest even (v_cvt_f16_f32 under the default rounding mode).  The empty asm pins
// the argument down as a float32 VALUE: without it the compiler folds a multiply or fma that produces f into
// v_fma_mixlo_f16, which rounds the exact product ONCE to float16, while the float32 paths of the reference
// round twice -- (q * scale) is a float32 tensor before .to(torch.float16) -- and the two differ on float16
// subnormal ties (0.234375 * 1.402537e-4 = 551.4999.. units of 2^-24: 551 fused, 552 in two steps).
__device__ __forceinline__ uint32_t f2h_bits(float f) {
  asm("" : "+v"(f));
  return h_bits((half_t)f);
}

// Deterministic sum over the 64 lanes of a wave (fixed butterfly order); every lane gets the result.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
