// Semantics probes for gfx950 instructions considered for the error accumulation:
//   v_dot2_f32_f16 / v_dot2c_f32_f16 on float16 subnormals (flushed or honoured?), rounding of the two-term sum,
//   v_fma_mix_f32 on float16 subnormals, v_max3_f16 with |.| modifiers on NaN inputs.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

__global__ void k_probe(const uint32_t* in, float* dot2, float* dot2c, float* mix, uint32_t* mx, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w = in[i];
  float acc = 0.f, r;
  asm volatile("v_dot2_f32_f16 %0, %1, %1, %2" : "=v"(r) : "v"(w), "v"(acc));
  dot2[i] = r;
  float c = 0.f;
  asm volatile("v_dot2c_f32_f16 %0, %1, %1" : "+v"(c) : "v"(w));
  dot2c[i] = c;
  float m = 0.f;
  // m = fma(|w.lo|, 1.0, 0) through the mixed-precision fma
  uint32_t one = 0x3c003c00u;
  asm volatile("v_fma_mix_f32 %0, |%1|, %2, %3 op_sel_hi:[1,1,0]" : "=v"(m) : "v"(w), "v"(one), "v"(acc));
  mix[i] = m;
  uint32_t z = 0, q;
  // packed maximum3 of (w, w swapped, 0): both halves hold max(lo, hi, 0) if NaN-free
  uint32_t sw = (w >> 16) | (w << 16);
  asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(q) : "v"(w), "v"(sw), "v"(z));
  mx[i] = q;
}

static float h2f(uint32_t h) {
  uint32_t e = (h >> 10) & 31u, m = h & 1023u;
  float v;
  if (e == 31) v = m ? NAN : INFINITY;
  else v = e == 0 ? ldexpf((float)m, -24) : ldexpf((float)(m | 1024u), (int)e - 25);
  return (h & 0x8000u) ? -v : v;
}

int main() {
  // inputs: lo = every float16 pattern, hi = a few partners
  const uint32_t partners[] = {0x0001, 0x03ff, 0x0400, 0x1234, 0x3c00, 0x8001, 0x7e00, 0x7c00, 0x2e66};
  const int np = sizeof(partners) / 4, n = 65536 * np;
  uint32_t* h = (uint32_t*)malloc(n * 4);
  for (int p = 0; p < np; ++p)
    for (int a = 0; a < 65536; ++a) h[p * 65536 + a] = (uint32_t)a | (partners[p] << 16);
  uint32_t *din, *dmx;
  float *d2, *d2c, *dm;
  hipMalloc(&din, n * 4); hipMalloc(&d2, n * 4); hipMalloc(&d2c, n * 4); hipMalloc(&dm, n * 4); hipMalloc(&dmx, n * 4);
  hipMemcpy(din, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_probe, dim3((n + 255) / 256), dim3(256), 0, 0, din, d2, d2c, dm, dmx, n);
  float *r2 = (float*)malloc(n * 4), *r2c = (float*)malloc(n * 4), *rm = (float*)malloc(n * 4);
  uint32_t* rx = (uint32_t*)malloc(n * 4);
  hipMemcpy(r2, d2, n * 4, hipMemcpyDeviceToHost); hipMemcpy(r2c, d2c, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(rm, dm, n * 4, hipMemcpyDeviceToHost); hipMemcpy(rx, dmx, n * 4, hipMemcpyDeviceToHost);
  long bad2 = 0, bad2c = 0, badm = 0, badx = 0, sub_bad2 = 0, sub_bad2c = 0, fin = 0;
  for (int i = 0; i < n; ++i) {
    uint32_t lo = h[i] & 0xffff, hi = h[i] >> 16;
    float a = h2f(lo), b = h2f(hi);
    bool finite = ((lo >> 10) & 31) != 31 && ((hi >> 10) & 31) != 31;
    if (finite) {
      ++fin;
      double ex = (double)a * a + (double)b * b;  // exact in double
      float want = (float)ex;                     // single rounding
      bool sub = ((lo >> 10) & 31) == 0 || ((hi >> 10) & 31) == 0;
      if (memcmp(&want, &r2[i], 4)) { ++bad2; if (sub) ++sub_bad2; if (bad2 <= 5) printf("dot2  lo %04x hi %04x got %a want %a\n", lo, hi, r2[i], want); }
      if (memcmp(&want, &r2c[i], 4)) { ++bad2c; if (sub) ++sub_bad2c; if (bad2c <= 5) printf("dot2c lo %04x hi %04x got %a want %a\n", lo, hi, r2c[i], want); }
      float wm = fabsf(a);
      if (memcmp(&wm, &rm[i], 4)) { ++badm; if (badm <= 5) printf("mix lo %04x got %a want %a\n", lo, rm[i], wm); }
    }
    // max3(|lo|, |hi|, 0): report the NaN behaviour
    bool nan_in = (((lo >> 10) & 31) == 31 && (lo & 1023)) || (((hi >> 10) & 31) == 31 && (hi & 1023));
    if (!nan_in) {
      float fa = h2f(lo), fb = h2f(hi);
      float fm = fa > fb ? fa : fb; if (!(fm > 0.f)) fm = 0.f;
      uint32_t want = 0; for (uint32_t c = 0; c < 0x7c01; ++c) if (h2f(c) == fm) { want = c; break; }
      if ((rx[i] & 0xffff) != want) { ++badx; if (badx <= 5) printf("max3 lo %04x hi %04x got %04x want %04x\n", lo, hi, rx[i] & 0xffff, want); }
    }
  }
  printf("finite pairs %ld: dot2 mismatches %ld (with a subnormal input %ld), dot2c %ld (%ld), mix-abs %ld, max3-abs(non-NaN) %ld\n",
         fin, bad2, sub_bad2, bad2c, sub_bad2c, badm, badx);
  // NaN behaviour samples
  for (int i = 0; i < n; ++i) {
    uint32_t lo = h[i] & 0xffff, hi = h[i] >> 16;
    if ((lo == 0x7e00 && hi == 0xbc00) || (lo == 0x3c00 && hi == 0x7d01) || (lo == 0x7e00 && hi == 0x3c00) || (lo == 0x3c00 && hi == 0x7e00) || (lo == 0x7d01 && hi == 0x3c00) || (lo == 0xfe00 && hi == 0x7c00))
      printf("pk_maximum3(%04x,%04x,0) = %04x   dot2 = %a\n", lo, hi, rx[i] & 0xffff, r2[i]);
  }
  return 0;
}
