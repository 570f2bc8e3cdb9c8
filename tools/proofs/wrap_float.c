/* Exhaustive check that the +-pi wrap of the WBFM demodulator (WbFmDemodulator.cc:417-425,
 * done in double by the reference) can be done with two float subtractions:
 *   reference:  (float)((double)d -+ 2*M_PI)          for |d| > M_PI
 *   float form: (d -+ C_HI) -+ C_LO,  C_HI = (float)(2*M_PI), C_LO = (float)(2*M_PI - C_HI)
 * for every d = fl(theta_a - theta_b) with theta_a, theta_b any two entries of the atan2
 * table (every phase difference the kernel can see).  d -+ C_HI is exact (Sterbenz), the second
 * subtraction rounds once; the check shows that rounding C_LO to float never changes it.
 * Build: gcc -O2 -ffp-contract=off -o wrap_float wrap_float.c -lm   (runs in under a minute) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

static int cmpf(const void *a, const void *b)
{
  const float x = *(const float *)a, y = *(const float *)b;
  return (x > y) - (x < y);
}

int main(void)
{
  static float t[65536];
  int n = 0;
  for (int q = -128; q < 128; q++)
    for (int i = -128; i < 128; i++)
      t[n++] = (float)atan2((double)q, (double)i);
  qsort(t, n, sizeof(float), cmpf);
  int m = 0;
  for (int k = 0; k < n; k++)
    if (m == 0 || t[k] != t[m - 1]) t[m++] = t[k];
  const float c_hi = (float)(2 * M_PI);
  const float c_lo = (float)(2 * M_PI - (double)c_hi);
  const float pi_up = 3.14159274101257324e+00f;       /* smallest float > M_PI */
  unsigned long long wraps = 0, bad = 0;
  for (int a = 0; a < m; a++)
  {
    for (int b = 0; b < m; b++)
    {
      volatile float d = t[a] - t[b];
      if (fabsf(d) < pi_up) continue;
      float ref = d;
      while (ref > M_PI) ref = (float)((double)ref - (2 * M_PI));
      while (ref < (-M_PI)) ref = (float)((double)ref + (2 * M_PI));
      volatile float u = (d > 0.0f) ? d - c_hi : d + c_hi;
      volatile float w = (d > 0.0f) ? u - c_lo : u + c_lo;
      wraps++;
      if (memcmp((const void *)&w, &ref, 4) != 0)
      {
        if (bad < 10) printf("MISMATCH d=%a ref=%a float=%a\n", d, ref, w);
        bad++;
      }
    }
  }
  printf("distinct thetas %d, wrapped pairs %llu, mismatches %llu, C_HI=%a C_LO=%a\n", m, wraps, bad, c_hi, c_lo);
  return bad != 0;
}
