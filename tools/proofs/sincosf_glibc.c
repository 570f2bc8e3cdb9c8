/* glibc 2.35 sinf / cosf (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h, s_sincosf_data.c: the ARM
 * optimized-routines algorithm: double-precision range reduction by pi/2 and two degree-7/8 polynomials), restated, and
 * checked against THIS host's libm over EVERY float of the argument range the device code uses (|x| < 120, the
 * reduce_fast branch: Nco::run passes phases in (-pi, pi], signals/fm.cc phases up to 2 pi).
 *
 * Why: Nco::run and the pm / fm generators call cos(float) / sin(float), which C++ overload resolution turns into
 * cosf / sinf.  libhrfd computed those in double and rounded (<= 1 ulp off: the last non-zero tolerance of
 * the repository).  With the algorithm restated the device produces libm's floats bit for bit.
 *
 * The x86-64 build of glibc dispatches between a plain and an -mfma -mavx2 build of the same source (ifunc): the
 * two can differ where a product-sum is contracted.  Both variants are evaluated here (FMA = 0 / 1) and the program
 * says which one -- or both -- equals the host's sinf / cosf everywhere.
 *
 * build: gcc -O2 -ffp-contract=off -fopenmp -o sincosf_glibc sincosf_glibc.c -lm      run: ./sincosf_glibc
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef struct
{
  double sign[4];
  double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3;
} sincos_t;

static const sincos_t T[2] = {
    {{1.0, -1.0, -1.0, 1.0},
     0x1.45F306DC9C883p+23,
     0x1.921FB54442D18p0,
     0x1p0,
     -0x1.ffffffd0c621cp-2,
     0x1.55553e1068f19p-5,
     -0x1.6c087e89a359dp-10,
     0x1.99343027bf8c3p-16,
     -0x1.555545995a603p-3,
     0x1.1107605230bc4p-7,
     -0x1.994eb3774cf24p-13},
    {{1.0, -1.0, -1.0, 1.0},
     0x1.45F306DC9C883p+23,
     0x1.921FB54442D18p0,
     -0x1p0,
     0x1.ffffffd0c621cp-2,
     -0x1.55553e1068f19p-5,
     0x1.6c087e89a359dp-10,
     -0x1.99343027bf8c3p-16,
     -0x1.555545995a603p-3,
     0x1.1107605230bc4p-7,
     -0x1.994eb3774cf24p-13}};

static inline uint32_t asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline uint32_t abstop12(float x) { return (asuint(x) >> 20) & 0x7ff; }

#define MA(FMA, a, b, c) ((FMA) ? fma((a), (b), (c)) : ((a) * (b) + (c)))

/* sinf_poly: n even -> sine polynomial of x, odd -> cosine polynomial */
static inline float poly(int FMA, double x, double x2, const sincos_t *p, int n)
{
  if ((n & 1) == 0)
  {
    const double x3 = x * x2;
    const double s1 = MA(FMA, x2, p->s3, p->s2);
    const double x7 = x3 * x2;
    const double s = MA(FMA, x3, p->s1, x);
    return (float)MA(FMA, x7, s1, s);
  }
  const double x4 = x2 * x2;
  const double c2 = MA(FMA, x2, p->c4, p->c3);
  const double c1 = MA(FMA, x2, p->c1, p->c0);
  const double x6 = x4 * x2;
  const double c = MA(FMA, x4, p->c2, c1);
  return (float)MA(FMA, x6, c2, c);
}

static inline double reduce_fast(int FMA, double x, const sincos_t *p, int *np)
{
  const double r = x * p->hpi_inv;
  const int n = ((int32_t)r + 0x800000) >> 24;
  *np = n;
  return FMA ? fma(-(double)n, p->hpi, x) : x - n * p->hpi;
}

float hrfd_sinf(int FMA, float y)
{
  double x = y;
  const sincos_t *p = &T[0];
  if (abstop12(y) < abstop12(0x1.921FB6p-1f))             /* |y| < pi/4 */
  {
    const double s = x * x;
    if (abstop12(y) < abstop12(0x1p-12f))
    {
      return y;
    }
    return poly(FMA, x, s, p, 0);
  }
  int n;
  x = reduce_fast(FMA, x, p, &n);
  const double s = p->sign[n & 3];
  if (n & 2)
  {
    p = &T[1];
  }
  return poly(FMA, x * s, x * x, p, n);
}

float hrfd_cosf(int FMA, float y)
{
  double x = y;
  const sincos_t *p = &T[0];
  if (abstop12(y) < abstop12(0x1.921FB6p-1f))
  {
    const double x2 = x * x;
    if (abstop12(y) < abstop12(0x1p-12f))
    {
      return 1.0f;
    }
    return poly(FMA, x, x2, p, 1);
  }
  int n;
  x = reduce_fast(FMA, x, p, &n);
  const double s = p->sign[n & 3];
  if (n & 2)
  {
    p = &T[1];
  }
  return poly(FMA, x * s, x * x, p, n ^ 1);
}

int main(void)
{
  /* every float with |x| < 120: bit patterns 0 .. bits(120.0f) - 1, both signs */
  const uint32_t top = asuint(120.0f);
  unsigned long long bad[2][2] = {{0, 0}, {0, 0}};
#pragma omp parallel for schedule(static) reduction(+ : bad)
  for (uint32_t b = 0; b < top; b++)
  {
    for (int sg = 0; sg < 2; sg++)
    {
      const uint32_t u = b | ((uint32_t)sg << 31);
      float x;
      memcpy(&x, &u, 4);
      const uint32_t ws = asuint(sinf(x)), wc = asuint(cosf(x));
      for (int f = 0; f < 2; f++)
      {
        bad[f][0] += asuint(hrfd_sinf(f, x)) != ws;
        bad[f][1] += asuint(hrfd_cosf(f, x)) != wc;
      }
    }
  }
  printf("floats checked: %llu (|x| < 120, both signs)\n", 2ull * top);
  for (int f = 0; f < 2; f++)
  {
    printf("variant %s: sinf mismatches %llu, cosf mismatches %llu\n", f ? "with fused multiply-adds (the -mfma build)" : "without contraction (plain build)   ",
           bad[f][0], bad[f][1]);
  }
  return (bad[0][0] + bad[0][1] == 0 || bad[1][0] + bad[1][1] == 0) ? 0 : 1;
}
