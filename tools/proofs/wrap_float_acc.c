/* Exhaustive check, for EVERY float a with pi < |a| <= 8, that the phase accumulator's wrap
 * (PhaseAccumulator.cc:166-176: `while (acc > M_PI) acc -= 2*M_PI;` evaluated in double, stored to
 * float) equals the two float subtractions (a -+ C_HI) -+ C_LO used by k_phase_scan when one
 * subtraction brings the value back into [-pi, pi].
 * Build: gcc -O2 -ffp-contract=off -o wrap_float_acc wrap_float_acc.c -lm */
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdint.h>

int main(void)
{
  const float c_hi = (float)(2 * M_PI);
  const float c_lo = (float)(2 * M_PI - (double)c_hi);
  unsigned long long n = 0, bad = 0, multi = 0;
  uint32_t lo, hi;
  float f = 3.14159274101257324e+00f;                    /* smallest float > M_PI */
  memcpy(&lo, &f, 4);
  f = 8.0f;
  memcpy(&hi, &f, 4);
  for (int sign = 0; sign < 2; sign++)
  {
    for (uint32_t b = lo; b <= hi; b++)
    {
      uint32_t bits = b | (sign ? 0x80000000u : 0u);
      float a;
      memcpy(&a, &bits, 4);
      float ref = a;
      int steps = 0;
      while (ref > M_PI) { ref = (float)((double)ref - (2 * M_PI)); steps++; }
      while (ref < (-M_PI)) { ref = (float)((double)ref + (2 * M_PI)); steps++; }
      if (steps != 1) { multi++; continue; }
      volatile float u = sign ? a + c_hi : a - c_hi;
      volatile float w = sign ? u + c_lo : u - c_lo;
      float wv = w;
      n++;
      if (memcmp(&wv, &ref, 4) != 0)
      {
        if (bad < 10) printf("MISMATCH a=%a ref=%a float=%a\n", a, ref, wv);
        bad++;
      }
    }
  }
  printf("single-wrap values checked %llu, mismatches %llu, values needing more than one wrap (left to the double path) %llu\n", n, bad, multi);
  return bad != 0;
}
