/* Exhaustive proof obligation for the kernel's division shortcut (mq_quant.hip, div_h):
 *   q0 = v * r;  q1 = fma(fma(-q0, s, v), r, q0)   with r = RN(1/s)
 * must equal the IEEE float32 quotient v / s for every finite float16 v >= 0 and every float16 s in
 * [0x00A8, 0x70E2] (the range torch.clamp(min=1e-5, max=1e4) leaves a float16 scale in); the sign of v
 * is symmetric.  Prints the number of mismatches of q1 (float32 level), of h(q1) and of h(q0).
 * Build: gcc -O2 -mf16c -mfma -fopenmp -ffp-contract=off div_exhaustive.c -lm */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
int main(void) {
  long bad_f32 = 0, bad_h = 0, bad_mul_only = 0, total = 0;
#pragma omp parallel for reduction(+ : bad_f32, bad_h, bad_mul_only, total) schedule(dynamic, 64)
  for (int sb = 0x00A8; sb <= 0x70E2; ++sb) {
    float s = _cvtsh_ss((unsigned short)sb), r = 1.0f / s;
    for (int vb = 0; vb < 0x7c00; ++vb) {
      float v = _cvtsh_ss((unsigned short)vb);
      float qt = v / s, q0 = v * r, q1 = fmaf(fmaf(-q0, s, v), r, q0);
      unsigned short ht = _cvtss_sh(qt, _MM_FROUND_TO_NEAREST_INT);
      bad_f32 += (q1 != qt);
      bad_h += (_cvtss_sh(q1, _MM_FROUND_TO_NEAREST_INT) != ht);
      bad_mul_only += (_cvtss_sh(q0, _MM_FROUND_TO_NEAREST_INT) != ht);
      ++total;
    }
  }
  printf("%ld %ld %ld %ld\n", total, bad_f32, bad_h, bad_mul_only);
  return 0;
}
