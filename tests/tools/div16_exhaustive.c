/* Exhaustive proof obligation for div_pk16 of mq_quant.hip: the float16 quotient x = h(v / s) of the statistics
 * kernels' integer formats, computed in float16 arithmetic alone:
 *     v' = v P,  q0 = h(v' r),  rem = fma16(-q0, s', v'),  q1 = fma16(rem, r, q0)
 * with P = 2^k the power of two that brings s' = s P into [1, 2) and r = RN16(1 / s').
 *
 * For EVERY float16 scale s in [2^-15, 0x70E2] (the upper clamp of the scale) and every finite float16 v >= 0
 * (the sequence is odd in v) with |v / s| <= 600:
 *   (a) rint(q1) == rint(h(v / s))                       -- all the integer formats take from the quotient
 *   (b) q1 has the bits of h(v / s) whenever v / s >= 2^-4
 * and, for each of the 1024 values of s', every float32 within 1 ulp of 1 / s' rounds to the same float16 r (the
 * kernel takes r from v_rcp_f32, whose error bound is 1 ulp; the closest call is s' = 0x1.ffcp+0, whose reciprocal
 * lies 2 float32 ulp above a float16 midpoint.  mq_selfcheck_div16 repeats the whole enumeration on the device with
 * the hardware's own reciprocals).
 *
 * float16 operations are emulated exactly: a product of two float16 numbers is exact in float32; fma16 forms
 * a * b + c exactly in double (the operands span fewer than 53 bits on this domain, asserted) and rounds once,
 * through a round-to-odd float32.  Prints "pairs bad_rint bad_bits bad_recip inexact_double bad_small"; all but the
 * first must be 0.
 * Build: gcc -O2 -mf16c -mfma -fopenmp -ffp-contract=off div16_exhaustive.c -lm */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline float h2f(unsigned b) { return _cvtsh_ss((unsigned short)b); }
static inline unsigned f2h(float f) { return _cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT); }
static inline float hr(float f) { return h2f(f2h(f)); }

/* double -> float32 with round-to-odd, then float32 -> float16 to nearest even: one correct rounding of the
 * double to float16 (float32 keeps more than two extra bits) */
static inline float round_odd_f32(double d) {
  float f = (float)d;
  if ((double)f == d) return f;
  uint32_t b;
  memcpy(&b, &f, 4);
  if (fabs((double)f) > fabs(d)) b -= 1; /* truncate toward zero */
  b |= 1u;
  memcpy(&f, &b, 4);
  return f;
}
/* fma of three float16 values (held as floats) rounded once to float16; *inexact counts sums a double cannot hold */
static inline float fma16(float a, float b, float c, long* inexact) {
  const double p = (double)a * (double)b; /* exact: 22 significant bits */
  const double s = p + (double)c;
  /* exactness of the double addition (two-sum error term) */
  const double bb = s - p, err = (p - (s - bb)) + ((double)c - bb);
  if (err != 0.0) ++*inexact;
  return hr(round_odd_f32(s));
}

/* argv: J QMAX (defaults 0 600).  J < 0 (-1, -2) is the SHARED power of two of the several-groupings kernel: a group
 * inside a row divides v 2^(k+J), the row's prescaled weights, so s' = s 2^(k+J) lies in [2^J, 2^(J+1)) and the bits are
 * exact from T = 2^(-4-J) <= 1/4 up (rint equal everywhere: what the integer formats need).  J > 0 is the table formats' variant: s' = s 2^k 2^J lies in [2^J, 2^(J+1)) (the
 * weight takes two power-of-two factors, P1 P2 = 2^(k+J), since 2^(k+J) itself may exceed float16), which moves the
 * exactness threshold down to T = 2^(-4-J); QMAX bounds the quotient.  Sixth number printed: quotients below T whose
 * q1 is not below T as well (must be 0: then every table whose smallest 0 / entry midpoint is >= T rounds q1 and
 * h(v / s) alike). */
int main(int argc, char** argv) {
  const unsigned S_LO = 0x0200u, S_HI = 0x70E2u;
  const int J = argc > 1 ? atoi(argv[1]) : 0;
  const float QMAX = argc > 2 ? (float)atof(argv[2]) : 600.0f;
  const float T = ldexpf(1.0f, -4 - J);
  long pairs = 0, bad_rint = 0, bad_bits = 0, bad_recip = 0, inexact = 0, bad_small = 0;
  for (unsigned m = 0; m < 1024; ++m) {
    const float s1 = h2f(0x3c00u | m);
    const double exact = 1.0 / (double)s1;
    float f = (float)exact;
    const unsigned want = f2h(f);
    if (f2h(nextafterf(f, 0.0f)) != want || f2h(nextafterf(f, 2.0f)) != want) ++bad_recip;
  }
#pragma omp parallel for reduction(+ : pairs, bad_rint, bad_bits, inexact, bad_small) schedule(dynamic, 64)
  for (unsigned sb = S_LO; sb <= S_HI; ++sb) {
    const float s = h2f(sb);
    int e;
    frexpf(s, &e);
    const int k = 1 - e;
    /* both factors are float16 numbers: k + J <= 15 + J; for J <= 0 there is ONE factor, as in the kernels */
    const int k1 = J > 0 ? (k + J + 1) / 2 : k + J, k2 = k + J - k1;
    const float P1 = ldexpf(1.0f, k1), P2 = ldexpf(1.0f, k2), s1 = ldexpf(s, k + J); /* exact */
    const float r = hr((float)(1.0 / (double)s1));
    for (unsigned vb = 0; vb < 0x7c00u; ++vb) {
      const float v = h2f(vb), want = hr(v / s);
      if (!(want <= QMAX)) break; /* the quotient grows with v */
      const float v1 = hr(hr(v * P1) * P2);
      const float q0 = hr(v1 * r);
      const float rem = fma16(-q0, s1, v1, &inexact);
      const float q1 = fma16(rem, r, q0, &inexact);
      ++pairs;
      if (rintf(q1) != rintf(want)) ++bad_rint;
      if (want >= T && f2h(q1) != f2h(want)) ++bad_bits;
      if (want < T && !(q1 < T)) ++bad_small;
    }
  }
  printf("%ld %ld %ld %ld %ld %ld\n", pairs, bad_rint, bad_bits, bad_recip, inexact, bad_small);
  return 0;
}
