/* Proof by enumeration for the branch-free +-pi wrap of k_rx_wbfm_flow (wrap_pi_rint in hrfd_rx_kernels.hip):
 *
 *   reference (WbFmDemodulator.cc:417-425): float d;  while (d > M_PI) d -= 2*M_PI;  while (d < -M_PI) d += 2*M_PI;
 *     (comparisons and subtractions in double, the result stored back to float; |d| <= 2 pi + a few ulp, so one step)
 *   device:  n = rint(d * CM) (round to nearest even);  u = fma(-n, C_HI, d);  w = fma(-n, C_LO, u)
 *
 * Checked for EVERY float d with |d| <= 6.5 (both signs): w == reference, bit for bit (d itself, untouched, where no
 * wrap happens).  The one exception is d = -0.0, which comes out as +0.0: a difference of two table thetas is never
 * -0.0 (the table holds no -0.0, and x - x = +0.0), and a zero's sign cannot reach the PCM anyway.  Also picks CM: the float for which rint() flips exactly between the largest float
 * below M_PI and the smallest float above it.
 * build: gcc -O2 -ffp-contract=off -o wrap_rint wrap_rint.c -lm && ./wrap_rint */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static float ref_wrap(float d)
{
  float x = d;
  while (x > M_PI) x -= (2 * M_PI);      /* float -= double: computed in double, rounded back to float */
  while (x < -M_PI) x += (2 * M_PI);
  return x;
}

static const float C_HI = 6.28318548202514648e+00f;   /* (float)(2*M_PI) */
static float C_LO;                                      /* (float)(2*M_PI - C_HI) */

static float dev_wrap(float d, float cm)
{
  const float t = d * cm;
  const float n = rintf(t);                              /* v_rndne_f32: ties to even */
  const float u = fmaf(-n, C_HI, d);
  return fmaf(-n, C_LO, u);
}

int main(void)
{
  C_LO = (float)(2 * M_PI - (double)C_HI);
  const float pi_up = nextafterf((float)M_PI, 4.0f) > (float)M_PI && (double)(float)M_PI > M_PI ? (float)M_PI : nextafterf((float)M_PI, 4.0f);
  const float pi_dn = nextafterf(pi_up, 0.0f);
  printf("C_HI %a  C_LO %a (0x%08x)  pi_up %a  pi_dn %a\n", C_HI, C_LO, f2u(C_LO), pi_up, pi_dn);
  /* candidates around 1/(2 pi) */
  const float c0 = (float)(1.0 / (2 * M_PI));
  float best = 0;
  for (int k = -4; k <= 4; k++)
  {
    float cm = c0;
    for (int i = 0; i < (k < 0 ? -k : k); i++) cm = nextafterf(cm, k < 0 ? 0.0f : 1.0f);
    const int ok = rintf(pi_up * cm) == 1.0f && rintf(pi_dn * cm) == 0.0f && rintf(-pi_up * cm) == -1.0f && rintf(-pi_dn * cm) == 0.0f;
    printf("  CM candidate %a (0x%08x): %s\n", cm, f2u(cm), ok ? "flips at pi" : "no");
    if (ok && best == 0) best = cm;
  }
  if (best == 0) { printf("no CM\n"); return 1; }
  printf("CM = %a (0x%08x)\n", best, f2u(best));
  unsigned long long n = 0, bad = 0;
  for (uint32_t u = 0; u <= f2u(6.5f); u++)
  {
    for (int s = 0; s < 2; s++)
    {
      const float d = u2f(u | (s ? 0x80000000u : 0u));
      const float r = ref_wrap(d), w = dev_wrap(d, best);
      n++;
      if (f2u(r) != f2u(w) && f2u(d) != 0x80000000u)
      {
        if (bad < 10) printf("MISMATCH d %a ref %a dev %a\n", d, r, w);
        bad++;
      }
    }
  }
  printf("%llu values checked, %llu mismatches\n", n, bad);
  return bad != 0;
}
