/* Exhaustive check, for EVERY float a with |a| <= 8, that the Nco phase accumulator's wrap
 * (PhaseAccumulator.cc:166-176: `while (acc > M_PI) acc -= 2*M_PI; while (acc < -M_PI) acc += 2*M_PI;`
 * evaluated in double, stored to float) equals the branch-free form k_phase_scan runs per step:
 *     k = rint(a * M)          M = 0x1.45f308p-3 (one ulp above (float)(1/(2 pi)): see below)
 *     u = fma(k, -C_HI, a)     C_HI = (float)(2 pi)
 *     w = fma(k, -C_LO, u)     C_LO = (float)(2 pi - C_HI)
 * i.e. k is +-1 exactly when the reference wraps (a >= the float above pi, or <= its negative) and 0
 * otherwise, one wrap always suffices in this range, and the two fused steps (k = +-1: each product is exact,
 * one rounding) round like the reference's double subtraction.
 * Build: gcc -O2 -ffp-contract=off -o wrap_rint_fma wrap_rint_fma.c -lm */
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdint.h>

int main(void)
{
  const float c_hi = (float)(2 * M_PI);
  const float c_lo = (float)(2 * M_PI - (double)c_hi);
  const float inv = (float)(1.0 / (2 * M_PI));
  const float M = nextafterf(inv, 1.0f);
  unsigned long long n = 0, bad = 0, wraps = 0;
  uint32_t hi;
  float f = 8.0f;
  memcpy(&hi, &f, 4);
  printf("M = %a (%.9g), C_HI = %a, C_LO = %a\n", M, M, c_hi, c_lo);
  for (int sign = 0; sign < 2; sign++)
  {
    for (uint32_t b = 0; b <= hi; b++)
    {
      uint32_t bits = b | (sign ? 0x80000000u : 0u);
      float a;
      memcpy(&a, &bits, 4);
      float ref = a;
      while (ref > M_PI) { ref = (float)((double)ref - (2 * M_PI)); }
      while (ref < (-M_PI)) { ref = (float)((double)ref + (2 * M_PI)); }
      volatile float t = a * M;
      const float k = rintf(t);                          /* round to nearest even, like v_rndne_f32 */
      const float u = fmaf(k, -c_hi, a);
      float w = fmaf(k, -c_lo, u);
      n++;
      wraps += (k != 0.0f);
      if (memcmp(&w, &ref, 4) != 0 && !(w == 0.0f && ref == 0.0f))
      {
        if (bad < 10) printf("MISMATCH a=%a ref=%a got=%a k=%g\n", a, ref, w, k);
        bad++;
      }
    }
  }
  printf("values checked %llu (of which wrapped %llu), mismatches %llu\n", n, wraps, bad);
  return bad != 0;
}
