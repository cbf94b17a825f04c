/* Exhaustive proof obligation for the statistics-only fast path of mq_quant.hip (quant8_stats): there the
 * error e = v - d is taken with ONE packed float16 subtraction and must equal the exact difference of the two
 * float16 numbers (what the reference computes in float32 from the two stored tensors).  Sterbenz: v - d is
 * representable when d = 0 or d / 2 <= v <= 2 d; this program does not argue, it enumerates every (scale, value,
 * dequantized value) combination the fast path admits and checks representability of v - d in float16.
 *
 *   part 1  d = h(rint(x) * s), x = h(v / s), for EVERY float16 scale s in [0x00A8, 0x70E2] and every finite
 *           float16 v >= 0 with |x| < 512 (odd-symmetric in v).  Covers symmetric INT of every bit width (its
 *           clamp to +-qmax never acts: |v / s| <= qmax (1 + 2^-10) on the fast path) and the asymmetric codes
 *           that the clamp to [0, qmax] leaves alone.
 *   part 2  asymmetric INT, bits 2 .. 8: every (rmin <= 0 <= rmax) pair of float16 numbers the fast path admits
 *           (finite range, scale below the upper clamp, |v / s| < 500) -> (s, zp); the values near rmax (resp.
 *           rmin) whose code the clamp cuts to qmax (resp. 0): d = h((qmax - zp) s) (resp. h(-zp s)).
 *   part 3  value tables (E, M, bias, with / without subnormals): every scale s = clamp(h(amax / max)) that is
 *           not cut by the upper clamp, every finite v of either sign with |v| <= the largest amax giving that
 *           scale; d = h(q s), q = the table entry with mid[i-1] < x <= mid[i].
 * Prints "part total bad" per line; every bad count must be 0.
 * Build: gcc -O2 -mf16c -mfma -fopenmp -ffp-contract=off sterbenz_exhaustive.c -lm */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

static inline float h2f(unsigned b) { return _cvtsh_ss((unsigned short)b); }
static inline unsigned f2h(float f) { return _cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT); }
static inline float hr(float f) { return h2f(f2h(f)); } /* one float16 rounding */
#define S_LO 0x00A8u
#define S_HI 0x70E2u

/* is v - d (two float16 values held as floats) a float16 number? */
static inline int exact16(float v, float d) {
  double diff = (double)v - (double)d;
  return (double)hr((float)diff) == diff && (double)(float)diff == diff;
}

static long part1(long* total) {
  long bad = 0, tot = 0;
#pragma omp parallel for reduction(+ : bad, tot) schedule(dynamic, 64)
  for (unsigned sb = S_LO; sb <= S_HI; ++sb) {
    const float s = h2f(sb);
    for (unsigned vb = 0; vb < 0x7c00u; ++vb) {
      const float v = h2f(vb), x = hr(v / s);
      if (!(x < 512.0f)) break; /* x grows with v */
      const float d = hr(rintf(x) * s);
      bad += !exact16(v, d);
      ++tot;
    }
  }
  *total = tot;
  return bad;
}

/* next float16 value above / below a finite one (bits), crossing zero through -0 / +0 */
static inline unsigned h_up(unsigned b) { return (b & 0x8000u) ? (b == 0x8000u ? 0x0000u : b - 1) : b + 1; }
static inline unsigned h_down(unsigned b) { return (b & 0x8000u) ? b + 1 : (b == 0x0000u ? 0x8000u : b - 1); }

static long part2(int bits, long* total) {
  const float qmax = (float)((1 << bits) - 1), lo = h2f(S_LO), hi = h2f(S_HI);
  long bad = 0, tot = 0;
#pragma omp parallel for reduction(+ : bad, tot) schedule(dynamic, 16)
  for (int mi = 0; mi <= 0x7c00; ++mi) { /* rmin = -(float16 with bits mi), mi == 0x7c00 stands for +0 */
    const unsigned minb = mi == 0x7c00 ? 0x0000u : (0x8000u | (unsigned)mi);
    const float rmin = h2f(minb);
    float last_s = -1.0f;
    for (int xb = 0x7bff; xb >= 0; --xb) { /* rmax descending: the first pair of each scale has the widest row */
      const float rmax = h2f((unsigned)xb);
      const float range = hr(rmax - rmin);
      if (!(range < 65504.0f)) continue;
      const float s0 = hr(range / qmax);
      if (s0 > hi) continue;
      const float s = s0 < lo ? lo : s0;
      if (s == last_s) continue;
      last_s = s;
      const float big = fmaxf(fabsf(rmin), fabsf(rmax));
      if (!(big / s < 500.001f)) continue;
      float zp = rintf(hr(-rmin / s));
      zp = zp < 0.0f ? 0.0f : (zp > qmax ? qmax : zp);
      const float kmax = qmax - zp, kmin = -zp;
      const float dmax = hr(kmax * s), dmin = hr(kmin * s);
      for (unsigned vb = (unsigned)xb;; vb = h_down(vb)) { /* from rmax down while the clamp to qmax acts */
        const float v = h2f(vb);
        if (!(rintf(hr(v / s)) > kmax)) break;
        bad += !exact16(v, dmax);
        ++tot;
        if (vb == minb) break;
      }
      for (unsigned vb = minb;; vb = h_up(vb)) { /* from rmin up while the clamp to 0 acts */
        const float v = h2f(vb);
        if (!(rintf(hr(v / s)) < kmin)) break;
        bad += !exact16(v, dmin);
        ++tot;
        if (vb == (unsigned)xb) break;
      }
      ++tot;
    }
  }
  *total = tot;
  return bad;
}

/* the value list of an (E, M, bias) table: quant_weight.py's closed forms (see include/mq.h, mq_format) */
static int table_values(int E, int M, int B, int sub, float* out) {
  int n = 0;
  float pos[512];
  int np = 0;
  if (sub) {
    for (int m = 1; m < (1 << M); ++m) pos[np++] = ldexpf((float)m, 1 - B - M);
    for (int e = 1; e < (1 << E); ++e)
      for (int m = 0; m < (1 << M); ++m) pos[np++] = ldexpf(1.0f + ldexpf((float)m, -M), e - B);
  } else {
    for (int e = 0; e < (1 << E); ++e)
      for (int m = 0; m < (1 << M); ++m)
        if (e || m) pos[np++] = ldexpf(1.0f + ldexpf((float)m, -M), e - B);
  }
  for (int i = np - 1; i >= 0; --i) out[n++] = -pos[i];
  out[n++] = 0.0f;
  for (int i = 0; i < np; ++i) out[n++] = pos[i];
  return n;
}

static long part3(int E, int M, int B, int sub, long* total) {
  static float vals[1100];
  const int n = table_values(E, M, B, sub, vals);
  const float tmax = vals[n - 1], lo = h2f(S_LO), hi = h2f(S_HI);
  /* q for every float16 x: the entry with mid[i-1] < x <= mid[i]; NaN never reaches the fast path */
  unsigned short* qtab = malloc(65536 * sizeof(unsigned short));
  for (unsigned xb = 0; xb < 65536; ++xb) {
    if ((xb & 0x7fffu) >= 0x7c00u) { qtab[xb] = 0; continue; }
    const float x = h2f(xb);
    int i = 0;
    while (i < n - 1 && (vals[i] + vals[i + 1]) * 0.5f < x) ++i; /* (midpoints of table values are exact floats) */
    qtab[xb] = (unsigned short)f2h(vals[i]);
  }
  /* largest amax (bits) that gives each scale without touching the upper clamp */
  int* amax_of = malloc((S_HI + 1) * sizeof(int));
  for (unsigned i = 0; i <= S_HI; ++i) amax_of[i] = -1;
  for (unsigned ab = 0; ab < 0x7c00u; ++ab) {
    const float s0 = hr(h2f(ab) / tmax);
    if (s0 > hi) continue;
    const unsigned sb = f2h(s0 < lo ? lo : s0);
    if ((int)ab > amax_of[sb]) amax_of[sb] = (int)ab;
  }
  long bad = 0, tot = 0;
#pragma omp parallel for reduction(+ : bad, tot) schedule(dynamic, 64)
  for (unsigned sb = S_LO; sb <= S_HI; ++sb) {
    if (amax_of[sb] < 0) continue;
    const float s = h2f(sb);
    for (unsigned vb = 0; vb <= (unsigned)amax_of[sb]; ++vb)
      for (unsigned sign = 0; sign < 2; ++sign) {
        const float v = h2f(vb | (sign << 15));
        const float q = h2f(qtab[f2h(v / s)]);
        bad += !exact16(v, hr(q * s));
        ++tot;
      }
  }
  free(qtab);
  free(amax_of);
  *total = tot;
  return bad;
}

int main(int argc, char** argv) {
  long tot, bad;
  bad = part1(&tot);
  printf("int_unclamped %ld %ld\n", tot, bad);
  for (int bits = 2; bits <= 8; ++bits) {
    bad = part2(bits, &tot);
    printf("asym_clamped_%d %ld %ld\n", bits, tot, bad);
    fflush(stdout);
  }
  /* name, E, M, bias, subnormals: quantization_amd/fused.py LUT_FORMATS */
  static const struct { const char* name; int E, M, B, sub; } T[] = {
      {"fp8_e4m3", 4, 3, 7, 0}, {"fp8_e3m4", 3, 4, 3, 0}, {"fp8_e2m5", 2, 5, 1, 0}, {"fp4", 2, 1, 0, 1},
      {"fp5_e2m2", 2, 2, -1, 1}, {"fp5_e3m1", 3, 1, 0, 1}, {"fp6_e2m3", 2, 3, -2, 1}, {"fp6_e3m2", 3, 2, -1, 1},
      {"int3", 0, 2, -1, 1}, {"int4", 0, 3, -2, 1}, {"int5", 0, 4, -3, 1}, {"int6", 0, 5, -4, 1},
      {"int8", 0, 7, -6, 1}};
  for (unsigned i = 0; i < sizeof(T) / sizeof(T[0]); ++i) {
    bad = part3(T[i].E, T[i].M, T[i].B, T[i].sub, &tot);
    printf("table_%s %ld %ld\n", T[i].name, tot, bad);
    fflush(stdout);
  }
  return 0;
}
