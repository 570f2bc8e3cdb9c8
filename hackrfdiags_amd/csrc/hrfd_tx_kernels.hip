// hackrfdiags_amd/csrc/hrfd_tx_kernels.hip -- gfx950 transmit kernels.
//
//   k_mod<SSB>     SsbModulator::acceptData (SsbModulator.cc:455-470):
//                  modulateSignal (:667-707): s = (int16)(pcm/2); I = delay line
//                  (16 taps {0 x15, 1.0}; 1.0 quantises to -32768: I[n] = -s[n-15]),
//                  Q = 31-tap Hilbert (negated for USB);
//                  increaseSampleRate (:499-619): eight x2 Q15 polyphase stages
//                  (Interpolator_int16.cc:398-418), (int8_t) narrowing, interleave.
//   k_mod<INTERP>  signals/interpolateSignal.cc:250-374: int16 IQ pairs through the
//                  same cascade with that tool's own (asymmetric) stage-1 table.
//
// Everything is integer FIR work (bit-exact), so blocks of one channel are
// independent given enough input history: a workgroup owns a tile of 64 input
// samples (32 KiB of output), re-derives the few history samples every stage
// needs from the PCM just before the tile (from the carried tail at the start
// of a call), keeps stages 1, 3, 4 and 5 in LDS (stage 2 lives in the registers of
// the thread that feeds stage 3 with it) and runs the last three x2 stages in
// registers so that every lane ends with 16 contiguous output bytes
// (8 IQ pairs): one coalesced 16-byte store per lane, write traffic only.
// Every tap is a literal (the loops are unrolled over compile-time tap indices: a
// run-time tap index costs a memory load per multiply), and the half-band stages use
// what their tables are: phase 1 is one tap of 16384, q15(16384 + 16384 x) = (x + 1) >> 1,
// and phase 0 is symmetric, h (a + d) + g (b + c).  From stage 2 on no value leaves the
// int16 range (the phases' sum of |taps| is < 1), so the reference's (int16_t)
// narrowing is the identity there; stages 0 and 1 narrow through their int16 stores.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hrfd {

constexpr int HRFD_MOD_RAILS = 100;     // internal kind: int16 (I,Q) rails in, modulator tables
constexpr int HRFD_MOD_WB_HEAD = 101;   // WBFM modulator: (pcm, 0) pairs in, rail 0 after stage 5 out (x32)
constexpr int HRFD_MOD_WB_TAIL = 102;   // WBFM modulator: 256 kS/s (I,Q) rails in, stages 6-8 (x8)
constexpr int HRFD_MOD_FM_PHASE = 103;  // FM modulator: the Nco PHASE of every 8 kS/s sample in (float), cos / sin -> rails in the stage-0 load (round 5)
#ifndef HRFD_MOD_TILE
#define HRFD_MOD_TILE 128               // round 6: 128 input samples per workgroup (64 until round 5): the stages' histories and the six barriers
#endif                                  // are paid once per 64 KiB of output instead of once per 32: -1.7 % on the SSB bank (profiles/r6_kmod_tile128_ab.txt)
#ifndef HRFD_MOD_ABLATE
#define HRFD_MOD_ABLATE 0
#endif
#ifndef HRFD_MOD_ZNUM
#define HRFD_MOD_ZNUM 3                  // round 6: the x8 tail takes two of its eight outputs per rail from stage 7's numerators (k_mod, `eight`)
#endif
constexpr int kModTile = HRFD_MOD_TILE;  // input samples per workgroup
#ifndef HRFD_MOD_THREADS
#define HRFD_MOD_THREADS 256
#endif
constexpr int kModThreads = HRFD_MOD_THREADS;
constexpr int kModTail = 64;            // carried input history per channel (>= 54)

struct ModParams
{
  const int16_t *in;        // SSB: [C][n] PCM; INTERP: [C][2n] IQ pairs
  int8_t *out;              // [C][512 n]
  const int16_t *tail_in;   // [C][4][kModTail]: rows 0,1 the stage-0 source (SSB: scaled PCM in row 0),
  int16_t *tail_out;        //   rows 2,3 the last kH0 samples of the I and Q rails as they were
                            //   produced (the Q rail carries the sideband sign of its time)
  const uint8_t *lsb;       // [C] sideband (SSB)
  uint32_t *wbstep;         // WB_HEAD: [C][32 n] out: the Nco step of every 256 kS/s sample (float bits)
  const float *param;       // WB_HEAD: [C] frequency deviation
  const uint32_t *wbtail;   // WB_TAIL: [C][2] the last two (I,Q) rail pairs of the previous call
  uint32_t n;               // input samples per channel
  uint32_t n_channels;
  int libm_fma;             // FM_PHASE: which build of glibc's sinf / cosf the host has (glibc_sinf)
  uint32_t tile0, tiles_launch;   // this launch covers tiles [tile0, tile0 + tiles_launch) of every channel (tiles_launch 0: all of
                                  // them) -- the WBFM modulator runs its passes in time slices beside the phase recurrence
};

// taps of the eight stages as the reference's constructors quantise them
__constant__ constexpr int16_t kS1Ssb[40] = {
    Q_AUDIO_D40[0],  Q_AUDIO_D40[1],  Q_AUDIO_D40[2],  Q_AUDIO_D40[3],  Q_AUDIO_D40[4],  Q_AUDIO_D40[5],
    Q_AUDIO_D40[6],  Q_AUDIO_D40[7],  Q_AUDIO_D40[8],  Q_AUDIO_D40[9],  Q_AUDIO_D40[10], Q_AUDIO_D40[11],
    Q_AUDIO_D40[12], Q_AUDIO_D40[13], Q_AUDIO_D40[14], Q_AUDIO_D40[15], Q_AUDIO_D40[16], Q_AUDIO_D40[17],
    Q_AUDIO_D40[18], Q_AUDIO_D40[19], Q_AUDIO_D40[20], Q_AUDIO_D40[21], Q_AUDIO_D40[22], Q_AUDIO_D40[23],
    Q_AUDIO_D40[24], Q_AUDIO_D40[25], Q_AUDIO_D40[26], Q_AUDIO_D40[27], Q_AUDIO_D40[28], Q_AUDIO_D40[29],
    Q_AUDIO_D40[30], Q_AUDIO_D40[31], Q_AUDIO_D40[32], Q_AUDIO_D40[33], Q_AUDIO_D40[34], Q_AUDIO_D40[35],
    Q_AUDIO_D40[36], Q_AUDIO_D40[37], Q_AUDIO_D40[38], Q_AUDIO_D40[39]};
__constant__ constexpr int16_t kS1Interp[40] = {
    Q_INTERPSIG_S1[0],  Q_INTERPSIG_S1[1],  Q_INTERPSIG_S1[2],  Q_INTERPSIG_S1[3],  Q_INTERPSIG_S1[4],
    Q_INTERPSIG_S1[5],  Q_INTERPSIG_S1[6],  Q_INTERPSIG_S1[7],  Q_INTERPSIG_S1[8],  Q_INTERPSIG_S1[9],
    Q_INTERPSIG_S1[10], Q_INTERPSIG_S1[11], Q_INTERPSIG_S1[12], Q_INTERPSIG_S1[13], Q_INTERPSIG_S1[14],
    Q_INTERPSIG_S1[15], Q_INTERPSIG_S1[16], Q_INTERPSIG_S1[17], Q_INTERPSIG_S1[18], Q_INTERPSIG_S1[19],
    Q_INTERPSIG_S1[20], Q_INTERPSIG_S1[21], Q_INTERPSIG_S1[22], Q_INTERPSIG_S1[23], Q_INTERPSIG_S1[24],
    Q_INTERPSIG_S1[25], Q_INTERPSIG_S1[26], Q_INTERPSIG_S1[27], Q_INTERPSIG_S1[28], Q_INTERPSIG_S1[29],
    Q_INTERPSIG_S1[30], Q_INTERPSIG_S1[31], Q_INTERPSIG_S1[32], Q_INTERPSIG_S1[33], Q_INTERPSIG_S1[34],
    Q_INTERPSIG_S1[35], Q_INTERPSIG_S1[36], Q_INTERPSIG_S1[37], Q_INTERPSIG_S1[38], Q_INTERPSIG_S1[39]};

// tap x sample (or sum of two samples): both fit 24 bits.  Written as the instruction: where only bits 15 .. 30 of a
// sum are looked at later, the compiler "simplifies" a negative tap to tap + 2^31, no longer sees a 24-bit operand and
// takes the full 32-bit multiply, which issues at a quarter of the rate.
template <int TAP>
__device__ __forceinline__ int mul24(const int x)
{
  static_assert(TAP >= -(1 << 23) && TAP < (1 << 23), "24-bit tap");
  int r;
  asm("v_mul_i32_i24_e32 %0, %1, %2" : "=v"(r) : "i"(TAP), "v"(x));
  return r;
}

// (the same with the tap as a uniform value: an unrolled loop's tap index is no template argument)
__device__ __forceinline__ int mul24s(const int tap, const int x)
{
  int r;
  asm("v_mul_i32_i24_e32 %0, %1, %2" : "=v"(r) : "s"(tap), "v"(x));
  return r;
}

// Q15 output of an interpolator phase: (16384 + sum) >> 15, low 16 bits
__device__ __forceinline__ int q15(int acc) { return (int)(short)(acc >> 15); }

// x2 stage with the 8-tap half-band prototype {-1445,0,9548,16384,9548,0,-1445,0} (INTERP_HB8), inputs
// x[n], x[n-1], x[n-2], x[n-3] = a, b, c, d: phase 0 taps (h0,h2,h4,h6) are symmetric, phase 1 is
// (0,16384,0,0): y1 = q15(16384 + 16384 b) = (b + 1) >> 1.
static_assert(Q_INTERP_HB8[0] == Q_INTERP_HB8[6] && Q_INTERP_HB8[2] == Q_INTERP_HB8[4] && Q_INTERP_HB8[3] == 16384 &&
              Q_INTERP_HB8[1] == 0 && Q_INTERP_HB8[5] == 0 && Q_INTERP_HB8[7] == 0, "hb8 relies on the table's shape");
__device__ __forceinline__ void hb8(int a, int b, int c, int d, int &y0, int &y1)
{
  y0 = ((1 << 14) + (int)Q_INTERP_HB8[0] * (a + d) + (int)Q_INTERP_HB8[2] * (b + c)) >> 15;
  y1 = (b + 1) >> 1;
}

// Two consecutive positions n (even) and n + 1 of that stage from three aligned dwords of its input,
// d0 = (x[n-4], x[n-3]), d1 = (x[n-2], x[n-1]), d2 = (x[n], x[n+1]): the outputs of n and of n + 1 as two packed
// pairs.  Phase 0 is h x[n-3] + g x[n-2] + g x[n-1] + h x[n]: two v_dot2_i32_i16 on sample pairs (v_alignbit shifts
// the pairs of the even position into place) instead of two sums and two multiplies in int32 -- for which the
// compiler chose a quarter-rate v_mul_lo_u32 where the tap is negative.
__device__ __forceinline__ void hb8_pair(const uint32_t d0, const uint32_t d1, const uint32_t d2, uint32_t &o0, uint32_t &o1)
{
  constexpr uint32_t h = (uint16_t)Q_INTERP_HB8[0], g = (uint16_t)Q_INTERP_HB8[2];
  constexpr uint32_t kHG = h | (g << 16), kGH = g | (h << 16);
  const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, 16);   // (x[n-3], x[n-2])
  const uint32_t hi = __builtin_amdgcn_alignbit(d2, d1, 16);   // (x[n-1], x[n])
  const int a0 = dot2(hi, kGH, dot2(lo, kHG, 1 << 14));
  const int a1 = dot2(d2, kGH, dot2(d1, kHG, 1 << 14));
  const int y1 = (((int)d1 >> 16) + 1) >> 1;                   // (x[n-1] + 1) >> 1
  const int y3 = ((int)(int16_t)d2 + 1) >> 1;                  // (x[n] + 1) >> 1
  o0 = ((uint32_t)(a0 >> 15) & 0xffffu) | ((uint32_t)y1 << 16);
  o1 = ((uint32_t)(a1 >> 15) & 0xffffu) | ((uint32_t)y3 << 16);
}

// x2 stage with a 4-tap prototype {h, 16384, h, 0}: phase 0 = h (x[n] + x[n-1]), phase 1 = (x[n] + 1) >> 1
template <int H>
__device__ __forceinline__ void hb4(int xn, int xm1, int &y0, int &y1)
{
  y0 = ((1 << 14) + H * (xn + xm1)) >> 15;
  y1 = (xn + 1) >> 1;
}
// (the same for inputs whose range the compiler cannot see -- values out of v_dot2: mul24)
template <int H>
__device__ __forceinline__ void hb4_m24(int xn, int xm1, int &y0, int &y1)
{
  y0 = ((1 << 14) + mul24<H>(xn + xm1)) >> 15;
  y1 = (xn + 1) >> 1;
}
// the same stage with its outputs as twice the Q15 numerators: z >> 16 is the output, byte 2 of z its low byte
template <int H>
__device__ __forceinline__ void hb4z(int xn, int xm1, int &z0, int &z1)
{
  z0 = (1 << 15) + 2 * H * (xn + xm1);
  z1 = (xn << 15) + (1 << 15);
}
static_assert(Q_INTERP_HB3[0] == Q_INTERP_HB3[2] && Q_INTERP_HB3[1] == 16384 && Q_INTERP_HB3[3] == 0, "hb4 relies on the table's shape");
static_assert(Q_INTERP_HB2[0] == Q_INTERP_HB2[2] && Q_INTERP_HB2[1] == 16384 && Q_INTERP_HB2[3] == 0, "hb4 relies on the table's shape");
static_assert(Q_INTERP_HB1[0] == Q_INTERP_HB1[2] && Q_INTERP_HB1[1] == 16384 && Q_INTERP_HB1[3] == 0, "hb4 relies on the table's shape");

// LDS layout per rail (int16), each stage with its history in front (stage 2 is never stored):
//   x0 [24 + T]   s1 [6 + 2T]   s3 [8 + 8T]   s4 [4 + 16T]   s5 [2 + 32T]
constexpr int kH0 = 24, kH1 = 6, kH3 = 8, kH4 = 4, kH5 = 2;
constexpr int kO0 = 0;
constexpr int kO1 = kO0 + kH0 + kModTile;
constexpr int kO3 = kO1 + kH1 + 2 * kModTile;
constexpr int kO4 = kO3 + kH3 + 8 * kModTile;
constexpr int kO5 = kO4 + kH4 + 16 * kModTile;
constexpr int kRail = kO5 + kH5 + 32 * kModTile + 6;

// (float)(p / d) for a constant d without the double division (tens of instructions in the per-sample passes): the
// product with the reciprocal is within two ulps (of double) of the correctly rounded quotient, so the float results
// can only differ when the product lies within a few ulps of a midpoint between two floats (its 29 dropped mantissa
// bits are 0x10000000 +- 8), or where the float is subnormal: those cases, and only those, take the division.
__device__ __forceinline__ float div_then_float(const double p, const double d, const double recip_d)
{
  const double q = p * recip_d;
  const uint64_t b = __builtin_bit_cast(uint64_t, q);
  const uint32_t low = (uint32_t)b & 0x1fffffffu;
  const uint32_t expo = (uint32_t)(b >> 52) & 0x7ffu;
  const bool risky = (low + 8u - 0x10000000u) <= 16u || expo < 1023u - 100u;
  return risky ? (float)(p / d) : (float)q;
}

// The items 0 .. LIMIT-1 of a pass dealt to the workgroup's threads: the whole rounds run without a predicate (their
// count is a constant: a loop "t = tid; t < LIMIT; t += threads" makes the compiler carry a per-lane trip count, an
// exec mask and a dozen instructions of loop control per round), the ragged last round under one compare.
template <int LIMIT, typename F>
__device__ __forceinline__ void wg_loop(const int tid, F &&body)
{
  constexpr int kWhole = LIMIT / kModThreads;
#pragma unroll
  for (int k = 0; k < kWhole; k++)
  {
    body(tid + k * kModThreads);
  }
  if constexpr (LIMIT % kModThreads != 0)
  {
    if (tid < LIMIT % kModThreads)
    {
      body(tid + kWhole * kModThreads);
    }
  }
}

// ---- glibc 2.35 sinf / cosf, restated (round 5) --------------------------------------------------------------
// The reference calls cos(float) / sin(float) under <math.h> + `using namespace std`: C++ overload resolution makes
// that cosf / sinf (Nco.cc:186-199, signals/pm.cc:41-53, fm.cc:44-77).  glibc's are the ARM
// optimized-routines algorithm (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h, s_sincosf_data.c): reduction by
// pi/2 in double (n = round(x * 2/pi) by an integer trick, x - n * pi/2), then one of two double polynomials, rounded
// to float once -- deterministic, so it can be the device's arithmetic as well: tools/proofs/sincosf_glibc.c checks
// this restatement against the host's libm on EVERY float with |x| < 120 (2.2e9 values): 0 mismatches with the
// fused multiply-adds of the -mfma build that x86-64 glibc dispatches to on an FMA-capable CPU, 34 (all at |x| > 17)
// without them.  FMA: which of the two the host's libm is -- probed by the host (libm_variant, hrfd_api_tx.hip).
// Outside the restated range (|x| >= 120, NaN) the double-precision cos / sin rounded to float stand in (never reached:
// every caller wraps its phase into (-2 pi, 2 pi)).
// Provenance: the eight polynomial coefficients and the reduction constants below are the published values of glibc
// 2.35's __sincosf_table (sysdeps/ieee754/flt-32/s_sincosf_data.c; ARM optimized-routines, dual-licensed MIT / Apache-2.0
// WITH LLVM-exception upstream, LGPL-2.1-or-later as shipped in glibc).  Third-party ARITHMETIC the reference depends on
// through libm, restated here because the device has no glibc; no code of the reference itself.
struct SinCosTab
{
  double c0, c1, c2, c3, c4, s1, s2, s3;
};
__device__ __forceinline__ SinCosTab sincos_tab(const bool flip)
{
  // __sincosf_table[0] and [1] (the second has the cosine polynomial negated)
  const double sg = flip ? -1.0 : 1.0;
  SinCosTab t;
  t.c0 = sg * 0x1p0;
  t.c1 = sg * -0x1.ffffffd0c621cp-2;
  t.c2 = sg * 0x1.55553e1068f19p-5;
  t.c3 = sg * -0x1.6c087e89a359dp-10;
  t.c4 = sg * 0x1.99343027bf8c3p-16;
  t.s1 = -0x1.555545995a603p-3;
  t.s2 = 0x1.1107605230bc4p-7;
  t.s3 = -0x1.994eb3774cf24p-13;
  return t;
}
template <bool FMA>
__device__ __forceinline__ double sc_ma(double a, double b, double c)
{
  if (FMA)
  {
    return __builtin_fma(a, b, c);
  }
  const double p = a * b;                                // (-ffp-contract=off: two roundings)
  return p + c;
}
// sinf_poly (sincosf.h): n even -> the sine polynomial of x, odd -> the cosine polynomial
template <bool FMA>
__device__ __forceinline__ float sc_poly(double x, double x2, const SinCosTab &p, int n)
{
  if ((n & 1) == 0)
  {
    const double x3 = x * x2;
    const double s1 = sc_ma<FMA>(x2, p.s3, p.s2);
    const double x7 = x3 * x2;
    const double s = sc_ma<FMA>(x3, p.s1, x);
    return (float)sc_ma<FMA>(x7, s1, s);
  }
  const double x4 = x2 * x2;
  const double c2 = sc_ma<FMA>(x2, p.c4, p.c3);
  const double c1 = sc_ma<FMA>(x2, p.c1, p.c0);
  const double x6 = x4 * x2;
  const double c = sc_ma<FMA>(x4, p.c2, c1);
  return (float)sc_ma<FMA>(x6, c2, c);
}
// COS = false: sinf(y); true: cosf(y)
template <bool FMA, bool COS>
__device__ __forceinline__ float glibc_sincosf_v(float y)
{
  const uint32_t top = (__builtin_bit_cast(uint32_t, y) >> 20) & 0x7ffu;    // abstop12
  double x = (double)y;
  if (top >= 0x42fu)                                     // |y| >= 120 (abstop12(120.0f)), infinity, NaN: outside the restated range
  {
    return COS ? (float)cos(x) : (float)sin(x);
  }
  if (top < 0x3f4u)                                      // |y| < pi / 4 (abstop12(0x1.921FB6p-1f))
  {
    if (top < 0x398u)                                    // |y| < 2^-12
    {
      return COS ? 1.0f : y;
    }
    return sc_poly<FMA>(x, x * x, sincos_tab(false), COS ? 1 : 0);
  }
  // reduce_fast: n = round(x * 2/pi) through r = x * (2/pi * 2^24), (int32)r + 2^23 >> 24
  const double r = x * 0x1.45F306DC9C883p+23;
  const int n = ((int)r + 0x800000) >> 24;
  const double hpi = 0x1.921FB54442D18p0;
  x = FMA ? __builtin_fma(-(double)n, hpi, x) : x - (double)n * hpi;
  const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;   // sign[] = {1, -1, -1, 1}
  return sc_poly<FMA>(x * s, x * x, sincos_tab((n & 2) != 0), COS ? (n ^ 1) : n);
}
// Both at once (the callers always want both): ONE range reduction and one x^2 -- the two polynomials are evaluated
// exactly as above (glibc's own sincosf_poly shares them the same way), so the pair equals (cosf(y), sinf(y)) bit for bit.
template <bool FMA>
__device__ __forceinline__ void glibc_sincosf_pair_v(float y, float &sn, float &cs)
{
  const uint32_t top = (__builtin_bit_cast(uint32_t, y) >> 20) & 0x7ffu;
  double x = (double)y;
  if (top >= 0x42fu)
  {
    sn = (float)sin(x);
    cs = (float)cos(x);
    return;
  }
  if (top < 0x3f4u)
  {
    if (top < 0x398u)
    {
      sn = y;
      cs = 1.0f;
      return;
    }
    const double x2 = x * x;
    const SinCosTab t = sincos_tab(false);
    sn = sc_poly<FMA>(x, x2, t, 0);
    cs = sc_poly<FMA>(x, x2, t, 1);
    return;
  }
  const double r = x * 0x1.45F306DC9C883p+23;
  const int n = ((int)r + 0x800000) >> 24;
  const double hpi = 0x1.921FB54442D18p0;
  x = FMA ? __builtin_fma(-(double)n, hpi, x) : x - (double)n * hpi;
  const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
  const SinCosTab t = sincos_tab((n & 2) != 0);
  const double xs = x * s, x2 = x * x;
  sn = sc_poly<FMA>(xs, x2, t, n);
  cs = sc_poly<FMA>(xs, x2, t, n ^ 1);
}
__device__ __forceinline__ void glibc_sincosf(float y, int fma_variant, float &sn, float &cs)
{
  if (fma_variant)
  {
    glibc_sincosf_pair_v<true>(y, sn, cs);
  }
  else
  {
    glibc_sincosf_pair_v<false>(y, sn, cs);
  }
}
__device__ __forceinline__ float glibc_sinf(float y, int fma_variant)
{
  return fma_variant ? glibc_sincosf_v<true, false>(y) : glibc_sincosf_v<false, false>(y);
}
__device__ __forceinline__ float glibc_cosf(float y, int fma_variant)
{
  return fma_variant ? glibc_sincosf_v<true, true>(y) : glibc_sincosf_v<false, true>(y);
}

// Stages 6, 7, 8 of the cascade (HB3, HB2, HB1: x8) for ONE 256 kS/s sample j of both rails, in registers: v[rail][0] =
// x5[j], v[rail][1] = x5[j - 1] (x5[j - 2] only reaches an output of sample j - 1) -> the sample's 8 output IQ pairs,
// 16 bytes.  Shared by k_mod (every kind: stage 5 in LDS) and k_wb_tail (the WBFM modulator's rails straight from the
// Nco lookup).  k8000: 1 << 15 in a scalar register (tail_k8000()).
__device__ __forceinline__ int tail_k8000()
{
  int k;                                                  // as a literal it doubles the size of every add that takes it
#if defined(HRFD_K8000_VGPR)
  asm("v_mov_b32 %0, 0x8000" : "=v"(k));                 // (A/B: a vector register instead of a scalar one)
#else
  asm("s_mov_b32 %0, 0x8000" : "=s"(k));
#endif
  return k;
}
__device__ __forceinline__ uint4 tail_eight(const int (&v)[2][2], const int k8000)
{
  (void)k8000;
  uint32_t w[4];
  int z[2][8];
#pragma unroll
  for (int rail = 0; rail < 2; rail++)
  {
    const int xa = v[rail][0], xb = v[rail][1];
    // stage 6 (HB3): y6[2j-1] (phase 1 of j-1), y6[2j], y6[2j+1]
    int a0, a1;
    const int p1 = (xb + 1) >> 1;                       // y6[2j-1], phase 1 of j - 1
    hb4<Q_INTERP_HB3[0]>(xa, xb, a0, a1);               // y6[2j], y6[2j+1]
    // stage 7 (HB2): y7[4j-1] (phase 1 of y6[2j-1]), y7[4j..4j+3]
    int b0, b1, b2, b3;
    const int q1 = (p1 + 1) >> 1;                       // y7[4j-1]
#if HRFD_MOD_ZNUM
    // (the numerators of b0 and b2 are kept: two of stage 8's outputs come straight from them, below; opaque to the
    //  compiler, which otherwise folds "n + 32768" back into a second multiply-add with another constant)
    int n0 = (1 << 14) + Q_INTERP_HB2[0] * (a0 + p1);
    int n2 = (1 << 14) + Q_INTERP_HB2[0] * (a1 + a0);
    asm("" : "+v"(n0));
    asm("" : "+v"(n2));
    b0 = n0 >> 15;                                      // y7[4j]
    b1 = (a0 + 1) >> 1;                                 // y7[4j+1]
    b2 = n2 >> 15;                                      // y7[4j+2]
    b3 = (a1 + 1) >> 1;                                 // y7[4j+3]
#else
    hb4<Q_INTERP_HB2[0]>(a0, p1, b0, b1);               // y7[4j], y7[4j+1]
    hb4<Q_INTERP_HB2[0]>(a1, a0, b2, b3);               // y7[4j+2], y7[4j+3]
#endif
    // stage 8 (HB1): y8[8j .. 8j+7], (int8_t) narrowing (:607-610).  Only the low byte of an output is kept, so the
    // outputs are left as TWICE their Q15 numerators -- the byte wanted is then byte 2 of the word, which v_perm picks
    // from two words at a time: no shift and no mask per output (hb4z)
#if HRFD_MOD_ZNUM >= 2
    // round 6, the same idea for the phase-1 inputs b1 and b3, themselves phase-1 outputs: with c1 = b1 + 1 = (a0 + 3) >> 1
    // (the same two instructions as b1) the output is byte 2 of c1 << 15 -- a plain shift where (b1 << 15) + (1 << 15) is a
    // 64-bit-encoded shift-add -- and the phase-0 outputs that take b1 take c1 with the 2 H subtracted from their constant
    constexpr int kH1x2 = 2 * Q_INTERP_HB1[0];
    const int c1 = (a0 + 3) >> 1, c3 = (a1 + 3) >> 1;   // b1 + 1, b3 + 1
    (void)b1;
    (void)b3;
    hb4z<Q_INTERP_HB1[0]>(b0, q1, z[rail][0], z[rail][1]);
    z[rail][2] = ((1 << 15) - kH1x2) + kH1x2 * (c1 + b0);
    z[rail][3] = c1 << 15;
    z[rail][4] = ((1 << 15) - kH1x2) + kH1x2 * (b2 + c1);
    z[rail][6] = ((1 << 15) - kH1x2) + kH1x2 * (c3 + b2);
    z[rail][7] = c3 << 15;
#else
    hb4z<Q_INTERP_HB1[0]>(b0, q1, z[rail][0], z[rail][1]);
    hb4z<Q_INTERP_HB1[0]>(b1, b0, z[rail][2], z[rail][3]);
    hb4z<Q_INTERP_HB1[0]>(b2, b1, z[rail][4], z[rail][5]);
    hb4z<Q_INTERP_HB1[0]>(b3, b2, z[rail][6], z[rail][7]);
#endif
#if HRFD_MOD_ZNUM
    // round 6: where the input of a phase-1 output is itself a phase-0 output, b = N >> 15, byte 2 of (b << 15) + (1 << 15)
    // is byte 2 of N + (1 << 15): the bits of N below 15 cannot carry into bit 15.  The compiler does not see that only
    // byte 2 is looked at and builds (N & 0xffff8000) + 0x8000 -- two instructions with 32-bit literals, four times per
    // sample -- where one add does.
    z[rail][1] = n0 + k8000;
    z[rail][5] = n2 + k8000;
#endif
  }
#pragma unroll
  for (int d = 0; d < 4; d++)
  {
    // output dword d = I[2d], Q[2d], I[2d+1], Q[2d+1]: byte 2 of z[0][2d], z[1][2d], z[0][2d+1], z[1][2d+1]
    const uint32_t lo = __builtin_amdgcn_perm((uint32_t)z[1][2 * d], (uint32_t)z[0][2 * d], 0x0c0c0602u);
    const uint32_t hi = __builtin_amdgcn_perm((uint32_t)z[1][2 * d + 1], (uint32_t)z[0][2 * d + 1], 0x06020c0cu);
    w[d] = lo | hi;
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

template <int KIND>
__global__ __launch_bounds__(kModThreads) void k_mod(const ModParams M)
{
  __shared__ int16_t src[2][kModTail + kModTile];         // stage-0 source with history
  __shared__ __attribute__((aligned(16))) int16_t r[2][kRail];   // the two rails, stages 0..5

  // INTERP and RAILS take int16 (I,Q) pairs; RAILS (the AM / FM modulators' baseband, produced by
  // k_am_rails / k_fm_rails) runs them through the modulators' stage-1 table, INTERP through
  // interpolateSignal's own
  constexpr bool kPairs = (KIND == HRFD_MOD_INTERP) || (KIND == HRFD_MOD_RAILS) || (KIND == HRFD_MOD_WB_HEAD) || (KIND == HRFD_MOD_FM_PHASE);
  constexpr bool kPhase = (KIND == HRFD_MOD_FM_PHASE);   // the input is [C][n] float phases: Nco::run's cosf / sinf, x 16000, (int16_t) here
                                                         // (FmModulator.cc:600-612; rounds 1-4: a pass of its own, k_fm_rails, on a second stream)
  constexpr bool kMono = (KIND == HRFD_MOD_WB_HEAD);      // the input is the PCM itself, [C][n]: rail 0, rail 1 is zero (WbFmModulator.cc:389-425)
#ifndef HRFD_MONO_ONE_RAIL
#define HRFD_MONO_ONE_RAIL 1
#endif
  // (round 6: stages 1-5 of the WBFM head pass run the ONE rail there is -- until round 5 half of their threads interpolated
  //  the zero rail, which nothing reads behind stage 0)
  constexpr int kRails = (kMono && HRFD_MONO_ONE_RAIL) ? 1 : 2;
  const uint32_t tiles = (M.n + kModTile - 1) / kModTile;
  const uint32_t tiles_l = (M.tiles_launch != 0u) ? M.tiles_launch : tiles;
  // Workgroup ids go round the eight XCDs, each with an L2 of its own: XCD x takes the channels x, x + 8, ... and a
  // channel's tiles one after the other, so that what an L2 writes back is one contiguous stream (32 KiB per
  // workgroup, consecutive workgroups of the XCD adjacent).  Measured on a pure store stream of this shape
  // (tools/ubench/store_shapes.hip): 5.75 TB/s with consecutive chunks going round the XCDs, 6.3-6.4 this way.
  const uint32_t xcd = blockIdx.x & 7u, bi = blockIdx.x >> 3;
  const uint32_t c = 8u * (bi / tiles_l) + xcd;
  const uint32_t tile = M.tile0 + (bi % tiles_l);
  if (c >= M.n_channels || tile >= tiles)
  {
    return;
  }
  const int tid = threadIdx.x;
  const int t0 = (int)tile * kModTile;                    // first input sample of the tile
  const int n = (int)M.n;
  if constexpr (KIND == HRFD_MOD_WB_TAIL)
  {
    // the rails arrive at 256 kS/s (k_wb_rails): they are stage 5's place in LDS; the two
    // samples in front of the tile come from the input or, at the start of a call, from the
    // previous call's last two pairs
    const uint32_t *rin = reinterpret_cast<const uint32_t *>(M.in) + (size_t)c * M.n * 32;
    const int n32 = (int)M.n * 32;
    wg_loop<kH5 + 32 * kModTile>(tid, [&](const int t)
    {
      const int g = 32 * t0 + t - kH5;
      uint32_t w = 0u;
      if (g < 0)
      {
        w = M.wbtail[(size_t)c * 2 + (2 + g)];
      }
      else if (g < n32)
      {
        w = rin[g];
      }
      r[0][kO5 + t] = (int16_t)(w & 0xffffu);
      r[1][kO5 + t] = (int16_t)(w >> 16);
    });
    __syncthreads();
  }
  else if (!(HRFD_MOD_ABLATE & 4))                        // (4: TIMING EXPERIMENT ONLY, stages 0 .. 5 skipped)
  {
    const int16_t *in = M.in + (size_t)c * M.n * ((kPairs && !kMono) ? 2 : 1);   // (kPhase: n floats per channel = 2 n int16)
    const int16_t *tin = M.tail_in + (size_t)c * 4 * kModTail;
    // kPhase: the rail pair of input sample g from its phase
    auto fm_pair = [&](const int g, int &a, int &b) {
      const float phase = reinterpret_cast<const float *>(in)[g];
      float iv, qv;
      glibc_sincosf(phase, M.libm_fma, qv, iv);
      iv = iv * 16000.0f;
      qv = qv * 16000.0f;
      a = (int)(short)(int)iv;
      b = (int)(short)(int)qv;
    };

    // ---- stage-0 source: scaled PCM (SSB) or the IQ pair (INTERP), history first
    wg_loop<kModTail + kModTile>(tid, [&](const int t)
    {
      const int g = t0 - kModTail + t;                      // global input index
      int a = 0, b = 0;
      if (g < 0)
      {
        a = tin[kModTail + g];
        b = tin[kModTail + kModTail + g];
      }
      else if (g < n)
      {
        if (kMono)
        {
          a = in[g];
        }
        else if (kPhase)
        {
          fm_pair(g, a, b);
        }
        else if (kPairs)
        {
          a = in[2 * g];
          b = in[2 * g + 1];
        }
        else
        {
          // scaledSample = (float)pcm / 2; (int16_t) truncates toward zero (:679-686)
          float f = (float)in[g];
          f = f / 2.0f;
          a = (int)f;
        }
      }
      src[0][t] = (int16_t)a;
      src[1][t] = (int16_t)b;
    });
    // the last tile of the call also leaves the new tail (the other buffer of the ping-pong)
    if (tile + 1 == tiles)
    {
      int16_t *tout = M.tail_out + (size_t)c * 4 * kModTail;
      for (int t = tid; t < kModTail; t += kModThreads)
      {
        const int g = n - kModTail + t;
        int a = 0, b = 0;
        if (g < 0)
        {
          const int o = kModTail + g;                       // still inside the old tail
          a = tin[o];
          b = tin[kModTail + o];
        }
        else if (kMono)
        {
          a = in[g];
        }
        else if (kPhase)
        {
          fm_pair(g, a, b);
        }
        else if (kPairs)
        {
          a = in[2 * g];
          b = in[2 * g + 1];
        }
        else
        {
          float f = (float)in[g];
          f = f / 2.0f;
          a = (int)f;
        }
        tout[t] = (int16_t)a;
        tout[kModTail + t] = (int16_t)b;
      }
    }
    __syncthreads();

    // ---- stage 0: the two rails at the input rate, x0[j] for j in [-kH0, kModTile)
    wg_loop<kH0 + kModTile>(tid, [&](const int t)
    {
      const int j = t - kH0;
      const int16_t *s0 = &src[0][kModTail + j];            // s[n], s0[-k] = s[n-k]
      int iv, qv;
      if (t0 + j < 0)
      {
        // before the call: the rails as the previous call produced them (a sideband
        // switch between calls must not re-sign samples already in the pipelines)
        iv = tin[2 * kModTail + kModTail + (t0 + j)];
        qv = tin[3 * kModTail + kModTail + (t0 + j)];
      }
      else if (kPairs)
      {
        iv = s0[0];
        qv = src[1][kModTail + j];
      }
      else
      {
        // delay line: Q15 tap -32768 at k = 15 (FirFilter_int16.cc:151-224)
        iv = q15((1 << 14) + (-32768) * (int)s0[-15]);
        // 31-tap Hilbert: the odd taps are zero and h[30 - k] = -h[k] (checked below), int32 wrap-around sum
        int acc = 1 << 14;
  #pragma unroll
        for (int k = 0; k < 15; k += 2)
        {
          static_assert(N_SSB_HILBERT == 31, "");
          acc += mul24s((int)Q_SSB_HILBERT[k], (int)s0[-k] - (int)s0[-(30 - k)]);
        }
        qv = q15(acc);
        if (!M.lsb[c])
        {
          qv = (int)(short)(-qv);                           // USB: qPhaseShifted = -qPhaseShifted (:696-699)
        }
      }
      r[0][kO0 + t] = (int16_t)iv;
      r[1][kO0 + t] = (int16_t)qv;
    });
    __syncthreads();
    if (tile + 1 == tiles)
    {
      // new rail tails: x0[g] for g in [n - kModTail, n); only the last kH0 are ever read
      int16_t *tout = M.tail_out + (size_t)c * 4 * kModTail;
      for (int t = tid; t < 2 * kModTail; t += kModThreads)
      {
        const int rail = t / kModTail, u = t - rail * kModTail;
        const int j = (n - t0) - kModTail + u;              // tile-relative index
        tout[(2 + rail) * kModTail + u] = (j >= -kH0) ? r[rail][kO0 + kH0 + j] : (int16_t)0;
      }
    }

    // ---- stage 1: 40-tap prototype, x2: both phases of one input position n per thread and rail,
    //      outputs m = 2n, 2n + 1 for n in [-kH1/2, tile).  Phase p is sum_j h[2j + p] x[n - j], j < 20: the twenty
    //      samples x[n-19 .. n] as ten int16 pairs (eleven aligned dwords, shifted by a sample where n is even) against
    //      the taps in pairs -- v_dot2_i32_i16, int32 wrap-around like the reference's accumulator.
    {
      constexpr const int16_t (&h)[40] = (KIND == HRFD_MOD_INTERP) ? Q_INTERPSIG_S1 : Q_AUDIO_D40;
      static_assert((kO0 + kH0 - kH1 / 2 - 19) >= 0 && (kRail % 2) == 0, "the dwords below");
      if (!(HRFD_MOD_ABLATE & 16))   // (TIMING EXPERIMENT ONLY when set)
      wg_loop<kRails * (kH1 / 2 + kModTile)>(tid, [&](const int t) {
        const int rail = (kRails == 2) ? (t & 1) : 0, u = (kRails == 2) ? (t >> 1) : t;
        const int first = kO0 + kH0 + (u - kH1 / 2) - 19;   // index of x[n - 19]
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&r[rail][first & ~1]);
        const uint32_t sh = (first & 1) ? 16u : 0u;
        int acc0 = 1 << 14, acc1 = 1 << 14;
  #pragma unroll
        for (int k = 0; k < 10; k++)
        {
          // (lo, hi) = x[n - 19 + 2k], x[n - 18 + 2k], i.e. x[n - j] for j = 19 - 2k and 18 - 2k
          const uint32_t pr = __builtin_amdgcn_alignbit(w[k + 1], w[k], sh);
          constexpr auto tap2 = [](int lo, int hi) { return ((uint32_t)(uint16_t)(int16_t)lo) | ((uint32_t)(uint16_t)(int16_t)hi << 16); };
          acc0 = dot2(pr, tap2(h[2 * (19 - 2 * k)], h[2 * (18 - 2 * k)]), acc0);
          acc1 = dot2(pr, tap2(h[2 * (19 - 2 * k) + 1], h[2 * (18 - 2 * k) + 1]), acc1);
        }
        // (int16 stores narrow: the 40-tap phases can exceed the int16 range for adversarial input, wrap is the contract)
        reinterpret_cast<uint32_t *>(&r[rail][kO1])[u] = ((uint32_t)(acc0 >> 15) & 0xffffu) | ((uint32_t)(acc1 >> 15) << 16);
      });
    }
    __syncthreads();
    // ---- stages 2 (HB8) and 3 (HB3): one stage-1 position p per thread and rail -> s2[2p], s2[2p+1] in
    //      registers (and s2[2p-1], the cheap phase-1 value of the position before) -> s3[4p .. 4p+3]
    if (!(HRFD_MOD_ABLATE & 32))   // (TIMING EXPERIMENT ONLY when set)
    wg_loop<kRails * (kH3 / 4 + 2 * kModTile)>(tid, [&](const int t)
    {
      const int rail = (kRails == 2) ? (t & 1) : 0, u = (kRails == 2) ? (t >> 1) : t;
      // stage-1 index p = u - kH3/4, from -2; x[p-3 .. p] as two sample pairs out of three aligned dwords (shifted by
      // a sample where p - 3 is odd), the HB8 phase 0 as two v_dot2 (hb8_pair has the arithmetic)
      constexpr int kFirst = kO1 + kH1 - kH3 / 4 - 3;      // index of x[p - 3] for u = 0
      static_assert(kFirst >= kO1 && (kO1 % 2) == 0, "inside stage 1's outputs");
      const int first = kFirst + u;
      const uint32_t *w = reinterpret_cast<const uint32_t *>(&r[rail][first & ~1]);
      const uint32_t sh = (first & 1) ? 16u : 0u;
      const uint32_t lo = __builtin_amdgcn_alignbit(w[1], w[0], sh);   // (x[p-3], x[p-2])
      const uint32_t hi = __builtin_amdgcn_alignbit(w[2], w[1], sh);   // (x[p-1], x[p])
      constexpr uint32_t th = (uint16_t)Q_INTERP_HB8[0], tg = (uint16_t)Q_INTERP_HB8[2];
      const int e0 = dot2(hi, tg | (th << 16), dot2(lo, th | (tg << 16), 1 << 14)) >> 15;   // s2[2p]
      const int e1 = ((int)(int16_t)hi + 1) >> 1;          // s2[2p+1] = (x[p-1] + 1) >> 1
      const int em1 = (((int)lo >> 16) + 1) >> 1;          // s2[2p-1] = (x[p-2] + 1) >> 1
      int y0, y1, y2, y3;
      hb4_m24<Q_INTERP_HB3[0]>(e0, em1, y0, y1);
      hb4_m24<Q_INTERP_HB3[0]>(e1, e0, y2, y3);
      // (int16 pairs: 4u is even)
      uint32_t *o = reinterpret_cast<uint32_t *>(&r[rail][kO3 + 4 * u]);
      o[0] = ((uint32_t)y0 & 0xffffu) | ((uint32_t)y1 << 16);
      o[1] = ((uint32_t)y2 & 0xffffu) | ((uint32_t)y3 << 16);
    });
    __syncthreads();
    // ---- stage 4: HB8, outputs m in [-kH4, 16*tile); two stage-3 positions per thread and rail
    if (!(HRFD_MOD_ABLATE & 64))   // (TIMING EXPERIMENT ONLY when set)
    wg_loop<kRails * ((kH4 / 2 + 8 * kModTile) / 2)>(tid, [&](const int t)
    {
      const int rail = (kRails == 2) ? (t & 1) : 0, u = (kRails == 2) ? (t >> 1) : t;   // u: the input positions n = 2u - kH4/2 and n + 1
      constexpr int kFirst = kO3 + kH3 - kH4 / 2 - 4;       // index of x[n - 4] for u = 0
      static_assert(kFirst >= kO3 && (kFirst % 2) == 0 && (kO4 % 2) == 0, "aligned dwords inside stage 3's outputs");
      const uint32_t *w = reinterpret_cast<const uint32_t *>(&r[rail][kFirst]) + u;
      uint32_t *o = reinterpret_cast<uint32_t *>(&r[rail][kO4 + 4 * u]);
      hb8_pair(w[0], w[1], w[2], o[0], o[1]);
    });
    __syncthreads();
    // ---- stage 5: HB8, outputs m in [-kH5, 32*tile)
    if (!(HRFD_MOD_ABLATE & 128))   // (TIMING EXPERIMENT ONLY when set)
    wg_loop<kRails * ((kH5 / 2 + 16 * kModTile + 1) / 2)>(tid, [&](const int t)
    {
      const int rail = (kRails == 2) ? (t & 1) : 0, u = (kRails == 2) ? (t >> 1) : t;   // the input positions n = 2u - kH5/2 - 1 (even) and n + 1
      constexpr int kFirst = kO4 + kH4 - kH5 / 2 - 1 - 4;   // index of x[n - 4] for u = 0: two samples in front of the stage's
      static_assert(kFirst >= 0 && (kFirst % 2) == 0 && ((kO5 + kH5) % 2) == 0, "aligned dwords in, aligned dwords out");   // history (they only reach the outputs of position -2, which are not kept)
      const uint32_t *w = reinterpret_cast<const uint32_t *>(&r[rail][kFirst]) + u;
      uint32_t o0, o1;
      hb8_pair(w[0], w[1], w[2], o0, o1);
      uint32_t *o = reinterpret_cast<uint32_t *>(&r[rail][kO5 + kH5 - 4]) + 2 * u;   // outputs 2n .. 2n + 3, n = 2u - 2
      if (u != 0)
      {
        o[0] = o0;
      }
      o[1] = o1;
    });
    __syncthreads();

    if constexpr (KIND == HRFD_MOD_WB_HEAD)
    {
      // WbFmModulator::increasePcmSampleRate ends here (WbFmModulator.cc:389-425) with rail 0 at 256 kS/s; what
      // leaves the kernel is the Nco step of every such sample, modulateSignal (:601-604): f = deviation * x / 1024
      // (float), step = (float)((2*M_PI*f)/256000) (double expression, PhaseAccumulator.cc:105)
      const int valid32 = 32 * min(kModTile, n - t0);
      uint32_t *cell = M.wbstep + ((size_t)c * M.n + t0) * 32;
      const float dev = M.param[c];
      const double two_pi = 6.283185307179586476925286766559;
      // (round 6: four consecutive cells per thread -- one 8-byte LDS read, one 16-byte store; until round 5 a dword each)
      static_assert(((kO5 + kH5) % 4) == 0, "8-byte reads of stage 5's outputs");
      for (int q = tid; q < valid32 / 4; q += kModThreads)
      {
        const uint2 x4 = *reinterpret_cast<const uint2 *>(&r[0][kO5 + kH5 + 4 * q]);
        const int xs[4] = {(int)(int16_t)(x4.x & 0xffffu), (int)x4.x >> 16, (int)(int16_t)(x4.y & 0xffffu), (int)x4.y >> 16};
        uint32_t st[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
          float f = dev * (float)xs[k];
          f = f / 1024.0f;
          st[k] = __builtin_bit_cast(uint32_t, div_then_float(two_pi * (double)f, 256000.0, 1.0 / 256000.0));
        }
        *reinterpret_cast<uint4 *>(cell + 4 * q) = make_uint4(st[0], st[1], st[2], st[3]);
      }
      return;
    }
  }

  // ---- stages 6, 7, 8 in registers: one 256 kS/s sample j -> 8 output IQ pairs
  const int valid = min(kModTile, n - t0);                // input samples really in this tile
  int8_t *out = M.out + ((size_t)c * M.n + t0) * 512;
  const int k8000 = tail_k8000();
  // the tail's inputs of sample j: x5[j], x5[j - 1] of both rails
  struct In3
  {
    int v[2][2];
  };
  auto fetch = [&](const int j) -> In3 {
    In3 f;
#pragma unroll
    for (int rail = 0; rail < 2; rail++)
    {
      const int16_t *x5 = &r[rail][kO5 + kH5];
      f.v[rail][0] = x5[j];
      f.v[rail][1] = x5[j - 1];
    }
    return f;
  };
  auto eight_of = [&](const In3 &f, const int j) -> uint4 {
#if (HRFD_MOD_ABLATE & 2)
    return make_uint4((uint32_t)j, (uint32_t)j * 3u, (uint32_t)tid, 7u);   // TIMING EXPERIMENT ONLY: no tail arithmetic
#endif
    (void)j;
    const int v[2][2] = {{f.v[0][0], f.v[0][1]}, {f.v[1][0], f.v[1][1]}};
    return tail_eight(v, k8000);
  };
  auto eight = [&](const int j) -> uint4 { return eight_of(fetch(j), j); };
  if (valid == kModTile)
  {
    // (a whole tile, the usual case: eight rounds of stores without a predicate)
#if (HRFD_MOD_ABLATE & 1)
    wg_loop<32 * kModTile>(tid, [&](const int j) {                        // TIMING EXPERIMENT ONLY: no stores
      const uint4 w4 = eight(j);
      if (w4.x == 0x12345678u && w4.y == 0x9abcdef0u && w4.z == 0x0fedcba9u)
      {
        *reinterpret_cast<uint4 *>(out + (size_t)j * 16) = w4;
      }
    });
#else
#if (HRFD_MOD_ABLATE & 8)
    wg_loop<32 * kModTile>(tid, [&](const int j) { *reinterpret_cast<uint4 *>(out + (size_t)j * 16) = eight(j); });   // TIMING EXPERIMENT ONLY: plain stores
#else
    // (nontemporal: the output is written once and read by nobody on this device -- 2.6 % on the whole kernel; sc0 / sc1
    //  beside or instead of nt: no better, alone 2-4 % worse)
#if HRFD_MOD_ZNUM >= 3
    // (round 6: the lane's byte offset is ONE register for all rounds and a round's 4 KiB step goes to the scalar base --
    //  written as out + 16 j the compiler rebuilds j and its shift in vector registers every round)
    {
      typedef unsigned int u4 __attribute__((ext_vector_type(4)));
      static_assert((32 * kModTile) % kModThreads == 0, "whole rounds");
      const uint32_t voff = (uint32_t)tid * 16u;
      // (the store is opaque to the compiler, which moves no LDS read across it: the next round's reads stand in front of it)
      In3 cur = fetch(tid);
#pragma unroll
      for (int k = 0; k < 32 * kModTile / kModThreads; k++)
      {
        In3 nxt = cur;
        if (k + 1 < 32 * kModTile / kModThreads)
        {
          nxt = fetch(tid + (k + 1) * kModThreads);
        }
        const uint4 w4 = eight_of(cur, tid + k * kModThreads);
        cur = nxt;
        int8_t *base = out + (size_t)k * (16 * kModThreads);   // uniform: a scalar register pair
        const u4 d4 = u4{w4.x, w4.y, w4.z, w4.w};
        // (s_nop 0: a store of more than 64 bits reads its data one cycle late -- a vector instruction right behind it must
        //  not write those registers; the compiler's hazard pass keeps that distance for its own stores and does not look
        //  into assembly.  Without it: wrong bytes in some launches, found by the digest of tools/mod_time.py.)
        asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 0" : : "v"(voff), "v"(d4), "s"(base));
      }
    }
#else
    wg_loop<32 * kModTile>(tid, [&](const int j) {
      typedef unsigned int u4 __attribute__((ext_vector_type(4)));
      const uint4 w4 = eight(j);
      __builtin_nontemporal_store(u4{w4.x, w4.y, w4.z, w4.w}, reinterpret_cast<u4 *>(out + (size_t)j * 16));
    });
#endif
#endif
#endif
  }
  else
  {
    wg_loop<32 * kModTile>(tid, [&](const int j) {
      const uint4 w4 = eight(j);
      if (j < 32 * valid)
      {
        *reinterpret_cast<uint4 *>(out + (size_t)j * 16) = w4;
      }
    });
  }
}

template __global__ void k_mod<HRFD_MOD_SSB>(const ModParams);
template __global__ void k_mod<HRFD_MOD_INTERP>(const ModParams);
template __global__ void k_mod<HRFD_MOD_RAILS>(const ModParams);
template __global__ void k_mod<HRFD_MOD_FM_PHASE>(const ModParams);
template __global__ void k_mod<HRFD_MOD_WB_HEAD>(const ModParams);
template __global__ void k_mod<HRFD_MOD_WB_TAIL>(const ModParams);

} // namespace hrfd

namespace hrfd {

// ---- AM / FM modulator basebands ---------------------------------------
// AmModulator::modulateSignal (AmModulator.cc:574-612): I = Q = (int16)(((pcm/32768)*m + 1)/2*128*250),
// float operations in that order.  One thread per sample; rails [C][2n] int16 (I,Q pairs).
struct BaseParams
{
  const int16_t *pcm;       // [C][n]
  int16_t *rails;           // [C][2n]
  const float *param;       // [C] modulation index (AM) / frequency deviation in Hz (FM)
  float *acc;               // [C] FM: Nco phase accumulator (persists across calls)
  float *phase;             // [C][n] FM scratch: phase of every sample
  uint32_t *wb;             // WBFM: [C][32 n] 4-byte cells: step -> phase -> (I,Q) rails, in place
  const float *cos_t, *sin_t; // WBFM: Nco::runFast tables (host libm, Nco.cc:50-61)
  const uint32_t *wbpack;   // WBFM: [16384] (int16)(cos_t[i] * 900) | (int16)(sin_t[i] * 900) << 16
  uint32_t *wbtail_out;     // WBFM: [C][2] the call's last two rail pairs (next call's history)
  uint32_t n, n_channels;
  uint32_t lo, len;         // k_wb_rails: input samples [lo, lo + len) of every channel (len 0: all): a time slice
  int libm_fma;             // which build of glibc's sinf / cosf the host has (glibc_sinf, above)
};

__global__ void k_am_rails(const BaseParams B)
{
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)B.n * B.n_channels)
  {
    return;
  }
  const uint32_t c = (uint32_t)(t / B.n);
  float signal = (float)B.pcm[t] / 32768.0f;
  signal = signal * B.param[c];
  signal = signal + 1.0f;
  signal = signal / 2.0f;
  signal = signal * 128.0f;
  signal = signal * 250.0f;
  int v = (int)signal;                                   // (int16_t) with x86 semantics: |signal| < 2^15 here
  v = (int)(short)v;
  reinterpret_cast<uint32_t *>(B.rails)[t] = ((uint32_t)v & 0xffffu) * 0x00010001u;
}

// The baseband generators of signals/ (am.cc:40-52, dsb.cc:38-46, pm.cc:41-53): one thread per
// sample, float operations in the reference's order ("*= 0.8" and "*= M_PI" are double
// multiplications rounded back to float; cos/sin of a float argument are the float overloads
// there, evaluated here in double and rounded: +-1 LSB).  Output: (I,Q) int16 pairs for k_mod<INTERP>.
template <int KIND>
__global__ void k_sig_rails(const BaseParams B)
{
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)B.n * B.n_channels)
  {
    return;
  }
  float s = (float)B.pcm[t];
  int vi, vq;
  if (KIND == HRFD_MOD_SIG_AM)
  {
    s = (float)((double)s * 0.8);
    s = s + 65536.0f;
    s = s / 4.0f;
    vi = vq = (int)s;
  }
  else if (KIND == HRFD_MOD_SIG_DSB)
  {
    s = s / 4.0f;
    vi = vq = (int)s;
  }
  else
  {
    s = s / 60000.0f;
    s = (float)((double)s * 3.14159265358979323846);
    float cs_, sn_;
    glibc_sincosf(s, B.libm_fma, sn_, cs_);                  // pm.cc:41-53: cos(float) is cosf
    const float ci = cs_ * 16000.0f;
    const float sq = sn_ * 16000.0f;
    vi = (int)ci;
    vq = (int)sq;
  }
  reinterpret_cast<uint32_t *>(B.rails)[t] = ((uint32_t)vi & 0xffffu) | ((uint32_t)vq << 16);
}

// signals/fm.cc:44-77: theta += (pcm / 65536) * 3.5 (float), wrapped into [-2pi, 2pi] with double
// compares and double subtractions stored back to float, then 16000*cos/sin.  The phase is a
// serial float recurrence per channel: one thread per channel (tooling, 8 kS/s).
__global__ void k_sig_fm(const BaseParams B)
{
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= B.n_channels)
  {
    return;
  }
  const double two_pi = 2 * 3.14159265358979323846;
  float theta = B.acc[c];
  const int16_t *in = B.pcm + (size_t)c * B.n;
  uint32_t *out = reinterpret_cast<uint32_t *>(B.rails) + (size_t)c * B.n;
  for (uint32_t k = 0; k < B.n; k++)
  {
    float tn = (float)in[k];
    tn = tn / 65536.0f;
    tn = tn * 3.5f;
    theta = theta + tn;
    while ((double)theta > two_pi)
    {
      theta = (float)((double)theta - two_pi);
    }
    while ((double)theta < -two_pi)
    {
      theta = (float)((double)theta + two_pi);
    }
    float cs_, sn_;
    glibc_sincosf(theta, B.libm_fma, sn_, cs_);
    const float ci = cs_ * 16000.0f;
    const float sq = sn_ * 16000.0f;
    out[k] = ((uint32_t)(int)ci & 0xffffu) | ((uint32_t)(int)sq << 16);
  }
  B.acc[c] = theta;
}

// FmModulator::modulateSignal (FmModulator.cc:586-627).  Pass 1, one thread per sample: the Nco
// step, f = deviation * pcm / 32768 (float), step = (float)((2*M_PI*f)/8000) (double expression,
// PhaseAccumulator.cc:95-107).  Pass 2 is k_phase_scan (the recurrence), pass 3 k_fm_rails.
__global__ void k_fm_step(const BaseParams B)
{
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)B.n * B.n_channels)
  {
    return;
  }
  const uint32_t c = (uint32_t)(t / B.n);
  const double two_pi = 6.283185307179586476925286766559;
  float f = B.param[c] * (float)B.pcm[t];
  f = f / 32768.0f;
  B.phase[t] = div_then_float(two_pi * (double)f, 8000.0, 1.0 / 8000.0);
}

// The Nco phase recurrence of both FM modulators (PhaseAccumulator.cc:157-181), step -> phase in place over `steps`
// 4-byte cells per channel: return the current phase, add the step in float, wrap with double compares and a double
// subtraction.  A float accumulate with a wrap is neither associative nor contracting (a start value that is off by
// one ulp stays off), so the recurrence is serial per channel: 64 channels per wave, and the time of a launch is
// steps x (instructions per step) x 4 cycles whatever the number of channels.  Everything else is kept out of that
// wave's instruction stream:
//   * the wrap is branch free, k = rint(acc * M) in {-1, 0, 1}, acc = fma(k, -C_LO, fma(k, -C_HI, acc)) -- equal to
//     the reference's loops for EVERY float |acc| <= 8 (tools/proofs/wrap_rint_fma.c: 2.2e9 values, M one ulp above
//     (float)(1/(2 pi)) so that the float above pi is the first to wrap): add, mul, rndne, fma, fma per step; a chunk
//     with a step above 4.85 (absurd deviations: the loaders look) or an accumulator above pi is redone with the loops;
//   * the cells move through LDS: loader waves bring chunks of 64 steps x 64 channels in with coalesced 16-byte
//     loads, storer waves take them out, the recurrence wave reads and writes its channel's row with ds_*_b128.
//     A wave that touches memory itself waits for it (one access per lane and 16 steps was a memory round trip per
//     16 steps: 60 ns per step).
// Flags in LDS, no barriers in the loop; every wait is bounded (an expired one aborts the workgroup and counts in *err).
constexpr int kPsChunk = 64;                              // steps per chunk
constexpr int kPsRow = kPsChunk + 4;                      // dwords per channel row in LDS: 16-byte aligned, rows 4 banks apart
#ifndef HRFD_PS_SLOTS
#define HRFD_PS_SLOTS 6
#endif
#ifndef HRFD_PS_LOADERS
#define HRFD_PS_LOADERS 3
#endif
#ifndef HRFD_PS_STORERS
#define HRFD_PS_STORERS 3
#endif
constexpr int kPsSlots = HRFD_PS_SLOTS;
constexpr int kPsLoaders = HRFD_PS_LOADERS, kPsStorers = HRFD_PS_STORERS;
constexpr int kPsThreads = 64 * (1 + kPsLoaders + kPsStorers);
#ifndef HRFD_PS_CWAVE
#define HRFD_PS_CWAVE 3
#endif
constexpr int kPsCompute = HRFD_PS_CWAVE;

__device__ __forceinline__ uint32_t ps_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void ps_st(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void ps_order() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// stress build -DHRFD_FLOW_CHAOS (not shipped): every mover and recurrence wave dawdles 0 .. ~14 us at random behind its hand-overs
__device__ __forceinline__ void ps_dawdle(const uint32_t seed)
{
#ifdef HRFD_FLOW_CHAOS
  uint32_t x = (uint32_t)blockIdx.x * 0x9E3779B1u ^ seed * 0x85EBCA77u ^ (uint32_t)__builtin_amdgcn_s_memrealtime();
  x ^= x >> 15;
  x *= 0x2C1B3C6Du;
  x ^= x >> 12;
  if ((x & 7u) == 0u)
  {
    for (uint32_t z = (x >> 8) & 7u; z != 0u; z--)
    {
      __builtin_amdgcn_s_sleep(64);
    }
  }
#else
  (void)seed;
#endif
}

// the reference's wrap (PhaseAccumulator.cc:166-176).  It does not terminate for an accumulator so large that
// subtracting 2 pi no longer changes it; this one gives up after 64 turns.
__device__ __forceinline__ float ps_wrap_loops(float acc)
{
  const double pi = 3.14159265358979323846, two_pi = 6.283185307179586476925286766559;
  for (int t = 0; t < 64 && (double)acc > pi; t++)
  {
    acc = (float)((double)acc - two_pi);
  }
  for (int t = 0; t < 64 && (double)acc < -pi; t++)
  {
    acc = (float)((double)acc + two_pi);
  }
  return acc;
}

// kPsChan: channels per workgroup (lanes of the recurrence wave in use).  The wave's time per step does not depend on
// it, so a bank that leaves CUs idle anyway is spread thinly (16 per workgroup: less LDS traffic beside the chain, 8 %).
template <int kPsChan>
__global__ __launch_bounds__(kPsThreads) void k_phase_scan(uint32_t *cells, size_t steps, size_t row_stride, float *acc_io, uint32_t n_channels, uint32_t *err)
{
  constexpr int kPsPieces = kPsChan / 4;                  // 16-byte pieces per mover lane and chunk
  __shared__ __attribute__((aligned(16))) uint32_t ring[kPsSlots][kPsChan * kPsRow];
  __shared__ uint32_t ready[kPsSlots];                    // slot holds chunk i, loaded: i + 1
  __shared__ uint32_t freed[kPsSlots];                    // slot held chunk j, stored out: j + 1
  __shared__ uint32_t ctl[2];                             // chunks computed; abort
  const int tid = threadIdx.x, lane = tid & 63;
  // roles: wave kPsCompute runs the recurrence (alone on its SIMD when waves go round the four SIMDs in order: 3 of 7),
  // then loaders, then storers
  const int hw_wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = (hw_wave == kPsCompute) ? 0 : (hw_wave < kPsCompute ? hw_wave + 1 : hw_wave);
  if (tid < kPsSlots)
  {
    ready[tid] = 0u;
    freed[tid] = 0u;
  }
  if (tid < 2)
  {
    ctl[tid] = 0u;
  }
  __syncthreads();
  const uint32_t c0 = blockIdx.x * (uint32_t)kPsChan;
  const uint32_t nchunks = (uint32_t)((steps + kPsChunk - 1) / kPsChunk);
  // bounded wait for `cond()`; false: expired or aborted
  auto wait_for = [&](auto cond) -> bool {
    for (uint32_t spins = 0; spins < (1u << 24); spins++)
    {
      if (cond())
      {
        ps_order();
        return true;
      }
      if (ps_ld(&ctl[1]) != 0u)
      {
        return false;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (lane == 0)
    {
      ps_st(&ctl[1], 1u);
      atomicAdd(err, 1u);
    }
    return false;
  };
  // a mover lane's share of a chunk: piece r = channels 4r .. 4r+3, this lane: channel 4r + lane/16, steps 4 (lane%16) .. +3
  const int mq = lane & 15, mc = lane >> 4;

  if (wave == 0)
  {
    // ------------------------------------------------------------ the recurrence: lane = channel
    // Per chunk: the next chunk's row is requested first (into the other register set), the previous chunk's
    // "done" goes out once its writes have landed (a counted wait: only the requests just issued stay open), then
    // the 64 steps of this chunk run out of registers and the row is written back.  No LDS latency is exposed in
    // the steady state, and the readiness of the loaders' slots is polled for several chunks at a time.
    __builtin_amdgcn_s_setprio(3);
    const bool mine = lane < kPsChan;
    const uint32_t c = c0 + (uint32_t)lane;
    float acc = (mine && c < n_channels) ? acc_io[c] : 0.0f;
    const float kM = 0x1.45f308p-3f, kChi = 0x1.921fb6p+2f, kClo = -0x1.777a5cp-23f;
    // the branch-free wrap needs |acc + step| <= 8: |acc| <= pi (true behind every wrap) and |step| <= 4.85 (the loaders
    // look at the steps of a chunk and mark it); anything else takes the loops
    bool wild = __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(acc) <= 3.1415927f)) != 0ull;
#ifdef HRFD_PS_PROBE
    const unsigned long long ps_t0 = __builtin_readcyclecounter();
    unsigned long long ps_waited = 0;
#endif
    uint32_t known = 0;                                   // chunks [0, known) are in their slots
    auto ensure = [&](uint32_t need) -> bool {
      if (need < known)
      {
        return true;
      }
#ifdef HRFD_PS_PROBE
      const unsigned long long tw = __builtin_readcyclecounter();
#endif
      const bool okw = wait_for([&] {
        const uint32_t ch = known + (uint32_t)lane;       // lanes 0 .. kPsSlots-1: is chunk `ch` in its slot?
        const bool ok = lane < kPsSlots && ch < nchunks && (ps_ld(&ready[ch % kPsSlots]) & 0x7fffffffu) == ch + 1u;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
        known += (uint32_t)__builtin_ctzll(~m);           // the run of consecutive chunks from `known` on
        return need < known;
      });
#ifdef HRFD_PS_PROBE
      ps_waited += __builtin_readcyclecounter() - tw;
#endif
      return okw;
    };
    typedef uint4 Row[kPsChunk / 4];
    auto request = [&](Row &in, uint32_t &flag, uint32_t i) {
      const int slot = (int)(i % kPsSlots);
      const uint32_t *row = &ring[slot][(mine ? lane : 0) * kPsRow];
      if (mine)
      {
#pragma unroll
        for (int j = 0; j < kPsChunk / 4; j++)
        {
          in[j] = *reinterpret_cast<const uint4 *>(row + 4 * j);
        }
      }
      flag = ps_ld(&ready[slot]);
    };
    auto chunk = [&](const Row &in, const uint32_t flag, uint32_t i) {
      uint32_t *row = &ring[(int)(i % kPsSlots)][(mine ? lane : 0) * kPsRow];
      Row o;
      float a = acc;
      auto one = [&](uint32_t step_bits) -> uint32_t {
        const uint32_t phase_bits = __builtin_bit_cast(uint32_t, a);
        a = a + __builtin_bit_cast(float, step_bits);
        const float k = __builtin_rintf(a * kM);
        a = __builtin_fmaf(k, -kChi, a);
        a = __builtin_fmaf(k, -kClo, a);
        return phase_bits;
      };
#pragma unroll
      for (int j = 0; j < kPsChunk / 4; j++)
      {
        o[j].x = one(in[j].x);
        o[j].y = one(in[j].y);
        o[j].z = one(in[j].z);
        o[j].w = one(in[j].w);
      }
      if (wild || (flag >> 31) != 0u)
      {
        // a step or an accumulator outside the range the branch-free wrap is proven for: the chunk again, with the loops
        a = acc;
        for (int k = 0; k < kPsChunk && mine; k++)
        {
          const float st = __builtin_bit_cast(float, row[k]);
          row[k] = __builtin_bit_cast(uint32_t, a);
          a = ps_wrap_loops(a + st);
        }
        wild = __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(a) <= 3.1415927f)) != 0ull;
      }
      else if (mine)
      {
#pragma unroll
        for (int j = 0; j < kPsChunk / 4; j++)
        {
          *reinterpret_cast<uint4 *>(row + 4 * j) = o[j];
        }
      }
      acc = a;
    };
    // one iteration: request chunk i + 1, release chunk i - 1, run chunk i
    auto turn = [&](const Row &cur, const uint32_t cur_flag, Row &nxt, uint32_t &nxt_flag, uint32_t i) -> bool {
      const bool more = i + 1u < nchunks;
      if (more)
      {
        if (!ensure(i + 1u))
        {
          return false;
        }
        request(nxt, nxt_flag, i + 1u);
        // LDS operations complete in order: all but the 15 youngest are through, chunk i - 1's writes among them
        static_assert(kPsChunk / 4 + 1 > 15, "the counted wait below");
        asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
      }
      else
      {
        ps_order();
      }
      if (i > 0u && lane == 0)
      {
        ps_st(&ctl[0], i);
      }
      ps_dawdle(3u * i);
      chunk(cur, cur_flag, i);
      return true;
    };
    Row ra, rb;
    uint32_t fa = 0u, fb = 0u;
    bool alive = nchunks != 0u && ensure(0u);
    if (alive)
    {
      request(ra, fa, 0u);
    }
    uint32_t i = 0;
    while (alive && i < nchunks)
    {
      alive = turn(ra, fa, rb, fb, i);
      i++;
      if (alive && i < nchunks)
      {
        alive = turn(rb, fb, ra, fa, i);
        i++;
      }
    }
    if (alive)
    {
      ps_order();
      if (lane == 0)
      {
        ps_st(&ctl[0], nchunks);
      }
    }
    if (mine && c < n_channels && alive)
    {
      acc_io[c] = acc;
    }
#ifdef HRFD_PS_PROBE
    if (lane == 0)
    {
      atomicAdd(&err[1], (uint32_t)((__builtin_readcyclecounter() - ps_t0) >> 8));   // cycles / 256 of the recurrence wave
      atomicAdd(&err[2], (uint32_t)(ps_waited >> 8));                                  // of which in ensure()
    }
#endif
  }
  else if (wave <= kPsLoaders)
  {
    // ------------------------------------------------------------ loaders: chunk i of the 64 channels -> slot i % kPsSlots
    for (uint32_t i = (uint32_t)(wave - 1); i < nchunks; i += kPsLoaders)
    {
      const int slot = (int)(i % kPsSlots);
      if (i >= (uint32_t)kPsSlots && !wait_for([&] { return ps_ld(&freed[slot]) == i - kPsSlots + 1u; }))
      {
        break;
      }
      const size_t k0 = (size_t)i * kPsChunk + 4 * (size_t)mq;
      uint4 v[kPsPieces];
#pragma unroll
      for (int r = 0; r < kPsPieces; r++)
      {
        const uint32_t ch = c0 + 4u * r + (uint32_t)mc;
        v[r] = make_uint4(0u, 0u, 0u, 0u);               // a zero step leaves the accumulator alone (x + 0 = x, no wrap: |x| <= pi)
        if (ch < n_channels && k0 < steps)
        {
          v[r] = *reinterpret_cast<const uint4 *>(cells + (size_t)ch * row_stride + k0);
        }
      }
      uint32_t big = 0u;                                   // the largest |step| as bits (a NaN or an infinity is larger still)
#pragma unroll
      for (int r = 0; r < kPsPieces; r++)
      {
        *reinterpret_cast<uint4 *>(&ring[slot][(4 * r + mc) * kPsRow + 4 * mq]) = v[r];
        big = max(max(big, v[r].x & 0x7fffffffu), max(v[r].y & 0x7fffffffu, max(v[r].z & 0x7fffffffu, v[r].w & 0x7fffffffu)));
      }
      const bool marked = __builtin_amdgcn_ballot_w64(big > __builtin_bit_cast(uint32_t, 4.85f)) != 0ull;
      ps_order();
      if (lane == 0)
      {
        ps_st(&ready[slot], (i + 1u) | (marked ? 0x80000000u : 0u));
      }
      ps_dawdle(3u * i + 1u);
    }
  }
  else
  {
    // ------------------------------------------------------------ storers: finished chunk j -> memory, slot free
    for (uint32_t j = (uint32_t)(wave - 1 - kPsLoaders); j < nchunks; j += kPsStorers)
    {
      const int slot = (int)(j % kPsSlots);
      if (!wait_for([&] { return ps_ld(&ctl[0]) >= j + 1u; }))
      {
        break;
      }
      const size_t k0 = (size_t)j * kPsChunk + 4 * (size_t)mq;
      const uint32_t nlive = (k0 < steps) ? (n_channels - c0 + 3u - (uint32_t)mc) / 4u : 0u;   // pieces r with c0 + 4r + mc < n_channels
#pragma unroll
      for (int r = 0; r < kPsPieces; r++)
      {
        const uint4 t = *reinterpret_cast<const uint4 *>(&ring[slot][(4 * r + mc) * kPsRow + 4 * mq]);
        if ((uint32_t)r < nlive)
        {
          *reinterpret_cast<uint4 *>(cells + (size_t)(c0 + 4u * r + (uint32_t)mc) * row_stride + k0) = t;
        }
      }
      ps_order();                                          // the slot has been read (the stores may still be on their way)
      if (lane == 0)
      {
        ps_st(&freed[slot], j + 1u);
      }
      ps_dawdle(3u * j + 2u);
    }
  }
}

template __global__ void k_phase_scan<64>(uint32_t *, size_t, size_t, float *, uint32_t, uint32_t *);

// ---- round 4: the recurrence as a shift register across a row of lanes -------------------------------------------
// k_phase_scan's recurrence wave spends 27 of its 43 cycles per step on its five instructions (a lone wave issues one
// vector instruction per ~5.4 cycles) and the rest on carrying cells between LDS and its registers: lane = channel means
// every step's input and output is a dword of its own in every lane.  Here lane = TIME: a channel is a row of 16 lanes,
// lane j of the row holds steps 4 j .. 4 j + 3 of a chunk of 64 exactly as one coalesced 16-byte load delivers them, and
// the accumulator travels along the row through the DPP operand of the add itself:
//     x0 = row_shr:1(w3) + step0       (lane j takes w3 of lane j - 1; lane 0 has no source and KEEPS its x0)
//     w0 = wrap(x0);  w1 = wrap(w0 + step1);  w2 = wrap(w1 + step2);  w3 = wrap(w2 + step3)     (mul, rndne, fma, fma each)
// sixteen times.  Every lane executes every round; lane 0's x0 was set to (last chunk's final w3) + step0 by a row_ror:1
// add in front, so after round t lanes 0 .. t hold their final values -- a lane whose left neighbour is final recomputes
// the same four values again (each has a register of its own), the lanes to the right compute on values that are not
// final yet and are overwritten when their turn comes.  After sixteen rounds w0..w3 of lane j are the accumulators BEHIND
// steps 4 j .. 4 j + 3, i.e. the phases of cells 4 j + 1 .. 4 j + 4: stored one cell to the right, as one 16-byte store.
// Five vector instructions per step, a DPP hop (two wait states and a slower operand path) per four, and nothing else: no
// LDS, no mover waves, no flags, no waits that could expire.  Four channels per wave, four waves (one per SIMD) per
// workgroup.
constexpr int kPrThreads = 256;
constexpr int kPrChunk = 64;                              // steps per chunk and row

template <bool ROR>
__device__ __forceinline__ void pr_add(float &x, const float w, const float st)
{
  // (the two wait states a DPP read of a register needs behind the VALU write of it: the compiler does not look into
  //  inline assembly)
  if (ROR)
  {
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 row_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(x) : "v"(w), "v"(st));
  }
  else
  {
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(w), "v"(st));
  }
}

// The pipeline is ONE piece of assembly: the steps are requested eight chunks ahead straight into the registers they
// are consumed from, with COUNTED waits (memory operations of a wave complete in order on this part: when chunk i is due,
// the operations behind its request are the requests of the chunks up to i + 7 and the stores of the chunks since) --
// registers that have a load on its way must not be the compiler's to move, and left to itself it also waits for
// everything in flight before every chunk (round 1's 60 ns per step).  Scalar base + one constant 32-bit offset per lane:
// no vector address arithmetic (every vector instruction of a lone wave costs 5.4 cycles).
//   v[40:71]  eight slots of four        v[72:75] the chunk's steps    v76 .. v79 w0 .. w3    v80 x    v81 k    v82 x0
//   s40 chunks done   s41 refused   s[42:43] address   s44 chunks - 1   s45 4.85f   s[46:47] lanes 0 .. 14 of every row
//   s48 round counter   s[50:51] scratch   s[52:53] lane 15 of every row
// A chunk holding a step above 4.85 (or a NaN) is refused: the pipeline stops in front of it with its steps in v[72:75].
#define HRFD_PR_WRAP(x, dst) \
  "v_mul_f32 v81, 0x3e22f984, " x "\n" \
  "v_rndne_f32 v81, v81\n" \
  "v_fmamk_f32 " dst ", v81, 0xc0c90fdb, " x "\n" \
  "v_fmac_f32 " dst ", 0x343bbd2e, v81\n"
#define HRFD_PR_LANE \
  HRFD_PR_WRAP("v82", "v76") \
  "v_add_f32 v80, v76, v73\n" HRFD_PR_WRAP("v80", "v77") \
  "v_add_f32 v80, v77, v74\n" HRFD_PR_WRAP("v80", "v78") \
  "v_add_f32 v80, v78, v75\n" HRFD_PR_WRAP("v80", "v79")
// (s_nop 1: the two wait states a DPP read of a register needs behind the VALU write of it)
#define HRFD_PR_ROUND \
  "s_nop 1\n" \
  "v_add_f32_dpp v82, v79, v72 row_shr:1 row_mask:0xf bank_mask:0xf\n" HRFD_PR_LANE
#define HRFD_PR_ADDR(chunk_sgpr_or_const) \
  "s_min_u32 s42, " chunk_sgpr_or_const ", s44\n" \
  "s_lshl_b32 s42, s42, 8\n" \
  "s_add_u32 s42, %[b0], s42\n" \
  "s_addc_u32 s43, %[b1], 0\n"
#define HRFD_PR_TURN(slot, s0, s1, s2, s3, behind) \
  "s_cmp_ge_u32 s40, %[n]\n" \
  "s_cbranch_scc1 9f\n" \
  "s_waitcnt vmcnt(" behind ")\n" \
  "v_mov_b32 v72, " s0 "\n" \
  "v_mov_b32 v73, " s1 "\n" \
  "v_mov_b32 v74, " s2 "\n" \
  "v_mov_b32 v75, " s3 "\n" \
  "v_cmp_nle_f32_e64 vcc, |v72|, s45\n" \
  "v_cmp_nle_f32_e64 s[50:51], |v73|, s45\n" \
  "s_or_b64 vcc, vcc, s[50:51]\n" \
  "v_cmp_nle_f32_e64 s[50:51], |v74|, s45\n" \
  "s_or_b64 vcc, vcc, s[50:51]\n" \
  "v_cmp_nle_f32_e64 s[50:51], |v75|, s45\n" \
  "s_or_b64 vcc, vcc, s[50:51]\n" \
  "s_cbranch_vccz 1f\n" \
  "s_mov_b32 s41, 1\n" \
  "s_branch 9f\n" \
  "1:\n" \
  "s_add_u32 s42, s40, 8\n" HRFD_PR_ADDR("s42") "global_load_dwordx4 " slot ", %[voff], s[42:43]\n" \
  "v_add_f32_dpp v82, v79, v72 row_ror:1 row_mask:0xf bank_mask:0xf\n" HRFD_PR_LANE \
  "s_mov_b32 s48, 5\n" \
  "2:\n" \
  HRFD_PR_ROUND HRFD_PR_ROUND HRFD_PR_ROUND \
  "s_sub_u32 s48, s48, 1\n" \
  "s_cmp_lg_u32 s48, 0\n" \
  "s_cbranch_scc1 2b\n" \
  "s_lshl_b32 s42, s40, 8\n" \
  "s_add_u32 s42, %[b0], s42\n" \
  "s_addc_u32 s43, %[b1], 0\n" \
  "s_add_u32 s40, s40, 1\n" \
  "s_cmp_eq_u32 s40, %[n]\n" \
  "s_cbranch_scc1 3f\n" \
  "global_store_dwordx4 %[voff], v[76:79], s[42:43] offset:4\n" \
  "s_branch 4f\n" \
  "3:\n" \
  "s_mov_b64 exec, s[46:47]\n" \
  "global_store_dwordx4 %[voff], v[76:79], s[42:43] offset:4\n" \
  "s_mov_b64 exec, s[52:53]\n" \
  "global_store_dwordx3 %[voff], v[76:78], s[42:43] offset:4\n" \
  "s_mov_b64 exec, s[54:55]\n" \
  "4:\n"

// w: the accumulator in front of the run in (lane 15 of the row is the one that counts: the first add takes it into lane
// 0), the one behind the last chunk done out (again lane 15).  Returns the chunks done; refused: the next one holds a step
// the branch-free wrap is not proven for, `held` are its steps.  Every lane of the wave must be active.
__device__ __forceinline__ uint32_t pr_pipeline(float &w, float (&held)[4], bool &refused, const uint32_t voff, const uint32_t *wbase, const uint32_t nchunks)
{
  const uint64_t b = (uint64_t)(uintptr_t)wbase;
  const uint32_t b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b), b1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
  const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)nchunks);
  uint32_t done, flag;
  asm volatile(
      "s_mov_b64 s[54:55], exec\n"                          // (the last chunk's stores mask lanes: the entry mask comes back behind them)
      "s_mov_b32 s40, 0\n"
      "s_mov_b32 s41, 0\n"
      "s_sub_u32 s44, %[n], 1\n"
      "s_mov_b32 s45, 0x409b3333\n"
      "s_mov_b32 s46, 0x7fff7fff\n"
      "s_mov_b32 s47, 0x7fff7fff\n"
      "s_mov_b32 s52, 0x80008000\n"
      "s_mov_b32 s53, 0x80008000\n"
      "v_mov_b32 v79, %[w]\n"
      "v_mov_b32 v72, 0\n"
      "v_mov_b32 v73, 0\n"
      "v_mov_b32 v74, 0\n"
      "v_mov_b32 v75, 0\n"
      "s_nop 4\n"
      HRFD_PR_ADDR("0") "global_load_dwordx4 v[40:43], %[voff], s[42:43]\n"
      HRFD_PR_ADDR("1") "global_load_dwordx4 v[44:47], %[voff], s[42:43]\n"
      HRFD_PR_ADDR("2") "global_load_dwordx4 v[48:51], %[voff], s[42:43]\n"
      HRFD_PR_ADDR("3") "global_load_dwordx4 v[52:55], %[voff], s[42:43]\n"
      HRFD_PR_ADDR("4") "global_load_dwordx4 v[56:59], %[voff], s[42:43]\n"
      HRFD_PR_ADDR("5") "global_load_dwordx4 v[60:63], %[voff], s[42:43]\n"
      HRFD_PR_ADDR("6") "global_load_dwordx4 v[64:67], %[voff], s[42:43]\n"
      HRFD_PR_ADDR("7") "global_load_dwordx4 v[68:71], %[voff], s[42:43]\n"
      // the first eight chunks: behind the request of chunk d are 7 - d of the first requests, d later ones and d stores
      HRFD_PR_TURN("v[40:43]", "v40", "v41", "v42", "v43", "7") HRFD_PR_TURN("v[44:47]", "v44", "v45", "v46", "v47", "8")
      HRFD_PR_TURN("v[48:51]", "v48", "v49", "v50", "v51", "9") HRFD_PR_TURN("v[52:55]", "v52", "v53", "v54", "v55", "10")
      HRFD_PR_TURN("v[56:59]", "v56", "v57", "v58", "v59", "11") HRFD_PR_TURN("v[60:63]", "v60", "v61", "v62", "v63", "12")
      HRFD_PR_TURN("v[64:67]", "v64", "v65", "v66", "v67", "13") HRFD_PR_TURN("v[68:71]", "v68", "v69", "v70", "v71", "14")
      // ... then 7 requests and 8 stores
      "8:\n"
      HRFD_PR_TURN("v[40:43]", "v40", "v41", "v42", "v43", "15") HRFD_PR_TURN("v[44:47]", "v44", "v45", "v46", "v47", "15")
      HRFD_PR_TURN("v[48:51]", "v48", "v49", "v50", "v51", "15") HRFD_PR_TURN("v[52:55]", "v52", "v53", "v54", "v55", "15")
      HRFD_PR_TURN("v[56:59]", "v56", "v57", "v58", "v59", "15") HRFD_PR_TURN("v[60:63]", "v60", "v61", "v62", "v63", "15")
      HRFD_PR_TURN("v[64:67]", "v64", "v65", "v66", "v67", "15") HRFD_PR_TURN("v[68:71]", "v68", "v69", "v70", "v71", "15")
      "s_branch 8b\n"
      "9:\n"
      "s_waitcnt vmcnt(0)\n"                                // (requests behind the end repeat the last chunk: they land in the slots)
      "v_mov_b32 %[w], v79\n"
      "v_mov_b32 %[h0], v72\n"
      "v_mov_b32 %[h1], v73\n"
      "v_mov_b32 %[h2], v74\n"
      "v_mov_b32 %[h3], v75\n"
      "s_mov_b32 %[done], s40\n"
      "s_mov_b32 %[flag], s41\n"
      : [w] "+v"(w), [h0] "=&v"(held[0]), [h1] "=&v"(held[1]), [h2] "=&v"(held[2]), [h3] "=&v"(held[3]), [done] "=&s"(done), [flag] "=&s"(flag)
      : [voff] "v"(voff), [b0] "s"(b0), [b1] "s"(b1), [n] "s"(n)
      : "memory", "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s50", "s51", "s52", "s53", "s54", "s55",
        "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59",
        "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",
        "v80", "v81", "v82");
  refused = flag != 0u;
  return done;
}
#undef HRFD_PR_TURN
#undef HRFD_PR_ADDR
#undef HRFD_PR_ROUND
#undef HRFD_PR_LANE
#undef HRFD_PR_WRAP

// steps: a multiple of 64, and 4 * row_stride cells must fit a 32-bit byte offset (the host sees to both).  row_stride
// need NOT be a multiple of four cells: the FM modulator's time slices pass the call's length (4163 in the tests), so a
// channel's row may start on any 4-byte boundary and the 16-byte loads and stores here (and the stores' offset:4) rely on
// gfx9's unaligned access mode for global memory (SH_MEM_CONFIG.alignment_mode = unaligned: what ROCm runs compute queues
// in).  tests/test_gpu_tx_nco.py::test_fm_modulator_time_slices keeps an odd stride with five channels.
// The call is a sequence of RUNS: the pipeline as far as it gets, then -- in front of a chunk it refuses, or while the
// accumulator is above pi -- one chunk with the reference's loops in the same arrangement, then the pipeline again.
// A run writes exactly its own cells: lane 15 of its last chunk keeps its last value (the accumulator for the run behind:
// the cell belongs to that run), and the run's first cell (whose step had to be read first) is written when the run is
// through.
__global__ __launch_bounds__(kPrThreads) void k_phase_rows(uint32_t *cells, size_t steps, size_t row_stride, float *acc_io, uint32_t n_channels)
{
  const int j = threadIdx.x & 15;
  // a row behind the bank repeats the bank's last channel (the same values to the same cells): no lane is ever masked
  // (rows of ONE wave run in lockstep, so the copies cannot disturb each other; a whole wave behind the bank leaves)
  const uint32_t cw = blockIdx.x * (uint32_t)(kPrThreads / 16) + 4u * (threadIdx.x >> 6);     // the wave's first channel
  const uint32_t nchunks = (uint32_t)(steps / kPrChunk);
  if (nchunks == 0u || cw >= n_channels)
  {
    return;
  }
  const uint32_t c = min(cw + ((threadIdx.x >> 4) & 3u), n_channels - 1u);
  const uint32_t *wbase = cells + (size_t)cw * row_stride;   // (uniform over the wave: pr_pipeline takes it into SGPRs)
  const uint32_t voff = (uint32_t)(((size_t)(c - cw) * row_stride + 4u * (size_t)j) * 4u);
  uint32_t *row = cells + (size_t)c * row_stride + 4u * (size_t)j;   // the lane's four cells of chunk 0
  float w = acc_io[c];                                    // every lane of the row; from the first chunk on, lane 15 is the one that counts
  uint32_t i = 0;                                         // chunks done
  bool wild = __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(w) <= 3.1415927f)) != 0ull;
  // the accumulator in front of a run, in lane 0 (whose cell it belongs in)
  auto carry_of = [&]() -> float {
    float cr;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(cr) : "v"(w));
    return cr;
  };
  // chunk i with the reference's loops (cur: its steps, taken before anything of it is stored)
  auto loops_chunk = [&](const float (&cur)[4]) {
    const float carry = carry_of();
    float x0, w0 = 0.0f, w1 = 0.0f, w2 = 0.0f;
    pr_add<true>(x0, w, cur[0]);
#pragma nounroll
    for (int t = 0; t < 16; t++)
    {
      if (t != 0)
      {
        pr_add<false>(x0, w, cur[0]);
      }
      w0 = ps_wrap_loops(x0);
      w1 = ps_wrap_loops(w0 + cur[1]);
      w2 = ps_wrap_loops(w1 + cur[2]);
      w = ps_wrap_loops(w2 + cur[3]);
    }
    uint32_t *cell = row + (size_t)kPrChunk * i;
    cell[1] = __builtin_bit_cast(uint32_t, w0);
    cell[2] = __builtin_bit_cast(uint32_t, w1);
    cell[3] = __builtin_bit_cast(uint32_t, w2);
    if (j != 15)
    {
      cell[4] = __builtin_bit_cast(uint32_t, w);
    }
    if (j == 0)
    {
      cell[0] = __builtin_bit_cast(uint32_t, carry);
    }
    i++;
    wild = __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(w) <= 3.1415927f)) != 0ull;
  };
#pragma nounroll
  while (i < nchunks)
  {
    if (wild)
    {
      const uint4 q = *reinterpret_cast<const uint4 *>(row + (size_t)kPrChunk * i);
      const float cur[4] = {__builtin_bit_cast(float, q.x), __builtin_bit_cast(float, q.y), __builtin_bit_cast(float, q.z), __builtin_bit_cast(float, q.w)};
      loops_chunk(cur);
      continue;
    }
    const float carry = carry_of();
    const uint32_t i0 = i;
    float held[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    bool refused = false;
    i += pr_pipeline(w, held, refused, voff, wbase + (size_t)kPrChunk * i0, nchunks - i0);
    if (i != i0 && j == 0)
    {
      row[(size_t)kPrChunk * i0] = __builtin_bit_cast(uint32_t, carry);
    }
    if (i < nchunks)
    {
      loops_chunk(held);                                  // (refused: a step above 4.85)
    }
  }
  if (j == 15)
  {
    acc_io[c] = w;
  }
}

// ---- round 6: the same arrangement with EIGHT steps per lane (chunks of 128 steps) ---------------------------------
// A row of 16 lanes covers 128 steps, lane j the steps 8j .. 8j+7 as TWO coalesced 16-byte loads deliver them: the DPP hop
// (two wait states and the slower operand path) comes once per eight steps instead of once per four, and the round's
// steps are read straight out of the slot registers the loads put them in -- k_phase_rows copies them first (four moves per
// chunk), because it reloads a slot in front of its rounds; here a slot is reloaded BEHIND them (seven chunks ahead instead
// of eight: ~10 us of latency cover at 12 ns per step).  tools/ubench/phase_scan_rate.hip priced the hop at 0.65 ns of
// the 12.15 ns a step takes; every caller whose step count is a multiple of 128 gets this kernel (the WBFM modulator's
// calls and time slices: multiples of 2048), the others k_phase_rows.
//   v[40:103] eight slots of eight     v112 .. v119 w0 .. w7     v120 x    v121 k    v122 x0    v[104:111] a refused chunk's steps
//   s40 chunks done   s41 refused   s[42:43] address   s44 chunks - 1   s45 4.85f   s[46:47] lanes 0 .. 14 of every row
//   s48 round counter   s[50:51] scratch   s[52:53] lane 15 of every row   s[54:55] the entry's exec mask
// Memory operations behind the loads of the chunk that is due (they complete in order): the first turns 14 + 2 d (the
// 7 - d chunks requested in front of the loop, two loads each, and two stores + two loads of every turn since), then 28.
constexpr int kPr8Chunk = 128;
#define HRFD_P8_WRAP(x, dst) \
  "v_mul_f32 v121, 0x3e22f984, " x "\n" \
  "v_rndne_f32 v121, v121\n" \
  "v_fmamk_f32 " dst ", v121, 0xc0c90fdb, " x "\n" \
  "v_fmac_f32 " dst ", 0x343bbd2e, v121\n"
#define HRFD_P8_LANE(s1, s2, s3, s4, s5, s6, s7) \
  HRFD_P8_WRAP("v122", "v112") \
  "v_add_f32 v120, v112, " s1 "\n" HRFD_P8_WRAP("v120", "v113") \
  "v_add_f32 v120, v113, " s2 "\n" HRFD_P8_WRAP("v120", "v114") \
  "v_add_f32 v120, v114, " s3 "\n" HRFD_P8_WRAP("v120", "v115") \
  "v_add_f32 v120, v115, " s4 "\n" HRFD_P8_WRAP("v120", "v116") \
  "v_add_f32 v120, v116, " s5 "\n" HRFD_P8_WRAP("v120", "v117") \
  "v_add_f32 v120, v117, " s6 "\n" HRFD_P8_WRAP("v120", "v118") \
  "v_add_f32 v120, v118, " s7 "\n" HRFD_P8_WRAP("v120", "v119")
#define HRFD_P8_ROUND(s0, s1, s2, s3, s4, s5, s6, s7) \
  "s_nop 1\n" \
  "v_add_f32_dpp v122, v119, " s0 " row_shr:1 row_mask:0xf bank_mask:0xf\n" HRFD_P8_LANE(s1, s2, s3, s4, s5, s6, s7)
#define HRFD_P8_ADDR(chunk_sgpr_or_const) \
  "s_min_u32 s42, " chunk_sgpr_or_const ", s44\n" \
  "s_lshl_b32 s42, s42, 9\n" \
  "s_add_u32 s42, %[b0], s42\n" \
  "s_addc_u32 s43, %[b1], 0\n"
#define HRFD_P8_CHECK(sa, sb) \
  "v_cmp_nle_f32_e64 s[50:51], |" sa "|, s45\n" \
  "s_or_b64 vcc, vcc, s[50:51]\n" \
  "v_cmp_nle_f32_e64 s[50:51], |" sb "|, s45\n" \
  "s_or_b64 vcc, vcc, s[50:51]\n"
#define HRFD_P8_TURN(lo, hi, s0, s1, s2, s3, s4, s5, s6, s7, behind) \
  "s_cmp_ge_u32 s40, %[n]\n" \
  "s_cbranch_scc1 9f\n" \
  "s_waitcnt vmcnt(" behind ")\n" \
  "s_mov_b64 vcc, 0\n" \
  HRFD_P8_CHECK(s0, s1) HRFD_P8_CHECK(s2, s3) HRFD_P8_CHECK(s4, s5) HRFD_P8_CHECK(s6, s7) \
  "s_cbranch_vccz 1f\n" \
  "v_mov_b32 v104, " s0 "\n" "v_mov_b32 v105, " s1 "\n" "v_mov_b32 v106, " s2 "\n" "v_mov_b32 v107, " s3 "\n" \
  "v_mov_b32 v108, " s4 "\n" "v_mov_b32 v109, " s5 "\n" "v_mov_b32 v110, " s6 "\n" "v_mov_b32 v111, " s7 "\n" \
  "s_mov_b32 s41, 1\n" \
  "s_branch 9f\n" \
  "1:\n" \
  "v_add_f32_dpp v122, v119, " s0 " row_ror:1 row_mask:0xf bank_mask:0xf\n" HRFD_P8_LANE(s1, s2, s3, s4, s5, s6, s7) \
  "s_mov_b32 s48, 5\n" \
  "2:\n" \
  HRFD_P8_ROUND(s0, s1, s2, s3, s4, s5, s6, s7) HRFD_P8_ROUND(s0, s1, s2, s3, s4, s5, s6, s7) HRFD_P8_ROUND(s0, s1, s2, s3, s4, s5, s6, s7) \
  "s_sub_u32 s48, s48, 1\n" \
  "s_cmp_lg_u32 s48, 0\n" \
  "s_cbranch_scc1 2b\n" \
  "s_lshl_b32 s42, s40, 9\n" \
  "s_add_u32 s42, %[b0], s42\n" \
  "s_addc_u32 s43, %[b1], 0\n" \
  "s_add_u32 s40, s40, 1\n" \
  "s_cmp_eq_u32 s40, %[n]\n" \
  "s_cbranch_scc1 3f\n" \
  "global_store_dwordx4 %[voff], v[112:115], s[42:43] offset:4\n" \
  "global_store_dwordx4 %[voff], v[116:119], s[42:43] offset:20\n" \
  "s_branch 4f\n" \
  "3:\n" \
  "s_mov_b64 exec, s[46:47]\n" \
  "global_store_dwordx4 %[voff], v[112:115], s[42:43] offset:4\n" \
  "global_store_dwordx4 %[voff], v[116:119], s[42:43] offset:20\n" \
  "s_mov_b64 exec, s[52:53]\n" \
  "global_store_dwordx4 %[voff], v[112:115], s[42:43] offset:4\n" \
  "global_store_dwordx3 %[voff], v[116:118], s[42:43] offset:20\n" \
  "s_mov_b64 exec, s[54:55]\n" \
  "4:\n" \
  "s_add_u32 s42, s40, 7\n" HRFD_P8_ADDR("s42") \
  "global_load_dwordx4 " lo ", %[voff], s[42:43]\n" \
  "global_load_dwordx4 " hi ", %[voff], s[42:43] offset:16\n"

// w: the accumulator in front of the run in (lane 15 of the row counts), the one behind the last chunk done out.  Returns
// the chunks done; refused: the next one holds a step the branch-free wrap is not proven for, `held` are its steps.
__device__ __forceinline__ uint32_t pr_pipeline8(float &w, float (&held)[8], bool &refused, const uint32_t voff, const uint32_t *wbase, const uint32_t nchunks)
{
  const uint64_t b = (uint64_t)(uintptr_t)wbase;
  const uint32_t b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b), b1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
  const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)nchunks);
  uint32_t done, flag;
#define HRFD_P8_T(A, B, C, D, E, F, G, H, behind) \
  HRFD_P8_TURN("v[" #A ":" #D "]", "v[" #E ":" #H "]", "v" #A, "v" #B, "v" #C, "v" #D, "v" #E, "v" #F, "v" #G, "v" #H, behind)
  asm volatile(
      "s_mov_b64 s[54:55], exec\n"
      "s_mov_b32 s40, 0\n"
      "s_mov_b32 s41, 0\n"
      "s_sub_u32 s44, %[n], 1\n"
      "s_mov_b32 s45, 0x409b3333\n"
      "s_mov_b32 s46, 0x7fff7fff\n"
      "s_mov_b32 s47, 0x7fff7fff\n"
      "s_mov_b32 s52, 0x80008000\n"
      "s_mov_b32 s53, 0x80008000\n"
      "v_mov_b32 v119, %[w]\n"
      "s_nop 4\n"
      HRFD_P8_ADDR("0") "global_load_dwordx4 v[40:43], %[voff], s[42:43]\n" "global_load_dwordx4 v[44:47], %[voff], s[42:43] offset:16\n"
      HRFD_P8_ADDR("1") "global_load_dwordx4 v[48:51], %[voff], s[42:43]\n" "global_load_dwordx4 v[52:55], %[voff], s[42:43] offset:16\n"
      HRFD_P8_ADDR("2") "global_load_dwordx4 v[56:59], %[voff], s[42:43]\n" "global_load_dwordx4 v[60:63], %[voff], s[42:43] offset:16\n"
      HRFD_P8_ADDR("3") "global_load_dwordx4 v[64:67], %[voff], s[42:43]\n" "global_load_dwordx4 v[68:71], %[voff], s[42:43] offset:16\n"
      HRFD_P8_ADDR("4") "global_load_dwordx4 v[72:75], %[voff], s[42:43]\n" "global_load_dwordx4 v[76:79], %[voff], s[42:43] offset:16\n"
      HRFD_P8_ADDR("5") "global_load_dwordx4 v[80:83], %[voff], s[42:43]\n" "global_load_dwordx4 v[84:87], %[voff], s[42:43] offset:16\n"
      HRFD_P8_ADDR("6") "global_load_dwordx4 v[88:91], %[voff], s[42:43]\n" "global_load_dwordx4 v[92:95], %[voff], s[42:43] offset:16\n"
      HRFD_P8_ADDR("7") "global_load_dwordx4 v[96:99], %[voff], s[42:43]\n" "global_load_dwordx4 v[100:103], %[voff], s[42:43] offset:16\n"
      HRFD_P8_T(40, 41, 42, 43, 44, 45, 46, 47, "14") HRFD_P8_T(48, 49, 50, 51, 52, 53, 54, 55, "16")
      HRFD_P8_T(56, 57, 58, 59, 60, 61, 62, 63, "18") HRFD_P8_T(64, 65, 66, 67, 68, 69, 70, 71, "20")
      HRFD_P8_T(72, 73, 74, 75, 76, 77, 78, 79, "22") HRFD_P8_T(80, 81, 82, 83, 84, 85, 86, 87, "24")
      HRFD_P8_T(88, 89, 90, 91, 92, 93, 94, 95, "26") HRFD_P8_T(96, 97, 98, 99, 100, 101, 102, 103, "28")
      "8:\n"
      HRFD_P8_T(40, 41, 42, 43, 44, 45, 46, 47, "28") HRFD_P8_T(48, 49, 50, 51, 52, 53, 54, 55, "28")
      HRFD_P8_T(56, 57, 58, 59, 60, 61, 62, 63, "28") HRFD_P8_T(64, 65, 66, 67, 68, 69, 70, 71, "28")
      HRFD_P8_T(72, 73, 74, 75, 76, 77, 78, 79, "28") HRFD_P8_T(80, 81, 82, 83, 84, 85, 86, 87, "28")
      HRFD_P8_T(88, 89, 90, 91, 92, 93, 94, 95, "28") HRFD_P8_T(96, 97, 98, 99, 100, 101, 102, 103, "28")
      "s_branch 8b\n"
      "9:\n"
      "s_waitcnt vmcnt(0)\n"                                // (requests behind the end repeat the last chunk: they land in the slots)
      "v_mov_b32 %[w], v119\n"
      "v_mov_b32 %[h0], v104\n"
      "v_mov_b32 %[h1], v105\n"
      "v_mov_b32 %[h2], v106\n"
      "v_mov_b32 %[h3], v107\n"
      "v_mov_b32 %[h4], v108\n"
      "v_mov_b32 %[h5], v109\n"
      "v_mov_b32 %[h6], v110\n"
      "v_mov_b32 %[h7], v111\n"
      "s_mov_b32 %[done], s40\n"
      "s_mov_b32 %[flag], s41\n"
      : [w] "+v"(w), [h0] "=&v"(held[0]), [h1] "=&v"(held[1]), [h2] "=&v"(held[2]), [h3] "=&v"(held[3]), [h4] "=&v"(held[4]),
        [h5] "=&v"(held[5]), [h6] "=&v"(held[6]), [h7] "=&v"(held[7]), [done] "=&s"(done), [flag] "=&s"(flag)
      : [voff] "v"(voff), [b0] "s"(b0), [b1] "s"(b1), [n] "s"(n)
      : "memory", "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s50", "s51", "s52", "s53", "s54", "s55",
        "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59",
        "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",
        "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99",
        "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116",
        "v117", "v118", "v119", "v120", "v121", "v122");
#undef HRFD_P8_T
  refused = flag != 0u;
  return done;
}
#undef HRFD_P8_TURN
#undef HRFD_P8_CHECK
#undef HRFD_P8_ADDR
#undef HRFD_P8_ROUND
#undef HRFD_P8_LANE
#undef HRFD_P8_WRAP

// steps: a multiple of 128.  Everything else as k_phase_rows (runs, refused chunks through the reference's loops, a run
// writes exactly its own cells).
__global__ __launch_bounds__(kPrThreads) void k_phase_rows8(uint32_t *cells, size_t steps, size_t row_stride, float *acc_io, uint32_t n_channels)
{
  const int j = threadIdx.x & 15;
  const uint32_t cw = blockIdx.x * (uint32_t)(kPrThreads / 16) + 4u * (threadIdx.x >> 6);     // the wave's first channel
  const uint32_t nchunks = (uint32_t)(steps / kPr8Chunk);
  if (nchunks == 0u || cw >= n_channels)
  {
    return;
  }
  const uint32_t c = min(cw + ((threadIdx.x >> 4) & 3u), n_channels - 1u);
  const uint32_t *wbase = cells + (size_t)cw * row_stride;
  const uint32_t voff = (uint32_t)(((size_t)(c - cw) * row_stride + 8u * (size_t)j) * 4u);
  uint32_t *row = cells + (size_t)c * row_stride + 8u * (size_t)j;   // the lane's eight cells of chunk 0
  float w = acc_io[c];
  uint32_t i = 0;
  bool wild = __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(w) <= 3.1415927f)) != 0ull;
  auto carry_of = [&]() -> float {
    float cr;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(cr) : "v"(w));
    return cr;
  };
  auto loops_chunk = [&](const float (&cur)[8]) {
    const float carry = carry_of();
    float x0, v[7] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    pr_add<true>(x0, w, cur[0]);
#pragma nounroll
    for (int t = 0; t < 16; t++)
    {
      if (t != 0)
      {
        pr_add<false>(x0, w, cur[0]);
      }
      v[0] = ps_wrap_loops(x0);
#pragma unroll
      for (int k = 1; k < 7; k++)
      {
        v[k] = ps_wrap_loops(v[k - 1] + cur[k]);
      }
      w = ps_wrap_loops(v[6] + cur[7]);
    }
    uint32_t *cell = row + (size_t)kPr8Chunk * i;
#pragma unroll
    for (int k = 0; k < 7; k++)
    {
      cell[1 + k] = __builtin_bit_cast(uint32_t, v[k]);
    }
    if (j != 15)
    {
      cell[8] = __builtin_bit_cast(uint32_t, w);
    }
    if (j == 0)
    {
      cell[0] = __builtin_bit_cast(uint32_t, carry);
    }
    i++;
    wild = __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(w) <= 3.1415927f)) != 0ull;
  };
#pragma nounroll
  while (i < nchunks)
  {
    if (wild)
    {
      const uint4 q0 = *reinterpret_cast<const uint4 *>(row + (size_t)kPr8Chunk * i), q1 = *reinterpret_cast<const uint4 *>(row + (size_t)kPr8Chunk * i + 4);
      const float cur[8] = {__builtin_bit_cast(float, q0.x), __builtin_bit_cast(float, q0.y), __builtin_bit_cast(float, q0.z), __builtin_bit_cast(float, q0.w),
                            __builtin_bit_cast(float, q1.x), __builtin_bit_cast(float, q1.y), __builtin_bit_cast(float, q1.z), __builtin_bit_cast(float, q1.w)};
      loops_chunk(cur);
      continue;
    }
    const float carry = carry_of();
    const uint32_t i0 = i;
    float held[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    bool refused = false;
    i += pr_pipeline8(w, held, refused, voff, wbase + (size_t)kPr8Chunk * i0, nchunks - i0);
    if (i != i0 && j == 0)
    {
      row[(size_t)kPr8Chunk * i0] = __builtin_bit_cast(uint32_t, carry);
    }
    if (i < nchunks)
    {
      loops_chunk(held);                                  // (refused: a step above 4.85)
    }
  }
  if (j == 15)
  {
    acc_io[c] = w;
  }
}

// the same recurrence for a cell count that is not a multiple of four (no 16-byte pieces): one thread per channel,
// straight from memory
__global__ void k_phase_scan_plain(uint32_t *cells, size_t steps, size_t row_stride, float *acc_io, uint32_t n_channels)
{
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_channels)
  {
    return;
  }
  float acc = acc_io[c];
  uint32_t *cell = cells + (size_t)c * row_stride;
  for (size_t k = 0; k < steps; k++)
  {
    const float st = __builtin_bit_cast(float, cell[k]);
    cell[k] = __builtin_bit_cast(uint32_t, acc);
    acc = ps_wrap_loops(acc + st);
  }
  acc_io[c] = acc;
}

// Pass 3, one thread per sample: Nco::run (Nco.cc:186-199) calls libm cosf/sinf -- glibc's algorithm restated on the
// device (glibc_cosf / glibc_sinf, above: bit for bit since round 5; rounds 1-4 rounded a double cos / sin and were
// within 1 LSB) --, times 16000, (int16_t).
__global__ void k_fm_rails(const BaseParams B)
{
  // (B.len != 0: the samples [lo, lo + len) of every channel -- a time slice of the call)
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t len = (B.len != 0u) ? B.len : B.n;
  if (t >= (size_t)len * B.n_channels)
  {
    return;
  }
  if (B.len != 0u)
  {
    const size_t c = t / len;
    t = c * B.n + B.lo + (t - c * len);
  }
  const float phase = B.phase[t];
  float iv, qv;
  glibc_sincosf(phase, B.libm_fma, qv, iv);
  iv = iv * 16000.0f;
  qv = qv * 16000.0f;
  const int i16 = (int)(short)(int)iv, q16 = (int)(short)(int)qv;
  reinterpret_cast<uint32_t *>(B.rails)[t] = ((uint32_t)i16 & 0xffffu) | ((uint32_t)q16 << 16);
}

// ---- WBFM modulator (WbFmModulator.cc:341-356) between the two halves of the cascade --------
// per sample, in parallel: Nco::runFast (Nco.cc:222-257) on the stored phase, x900, (int16_t):
// phase -> (I,Q) rail pair in place; the call's last two pairs are kept for the next call.
// The two float tables, x900 and narrowed, are one table of 16384 rail pairs (B.wbpack, built by the host with the
// same float multiply): 64 KiB, held in LDS by every workgroup -- one ds_read per sample instead of two gathers
// from memory -- and the workgroups walk the cells four at a time (16-byte accesses).
// Nco::runFast (Nco.cc:222-257) on a stored phase, x900, (int16_t): the rail pair of one 256 kS/s sample out of the
// packed table
__device__ __forceinline__ uint32_t wb_lookup(const uint32_t *pack, const uint32_t phase_bits)
{
  const double two_pi = 6.283185307179586476925286766559;
  const float scaled = __builtin_bit_cast(float, phase_bits) * 16384.0f;
  // (int)(x / two_pi): the product with the reciprocal is within two ulps of the quotient, so the truncation can only
  // differ when an integer is that close -- then, and only then, the division itself
  double quot = (double)scaled * (1.0 / two_pi);
  if (__builtin_fabs(quot - __builtin_rint(quot)) < 1e-6)
  {
    quot = (double)scaled / two_pi;
  }
  int idx = (int)(short)(int)quot;
  idx += 8192;
  idx = max(0, min(16383, idx));
  return pack[idx];
}

constexpr int kWbRailsThreads = 512;
__global__ __launch_bounds__(kWbRailsThreads) void k_wb_rails(const BaseParams B)
{
  __shared__ __attribute__((aligned(16))) uint32_t pack[16384];
  for (int i = threadIdx.x; i < 16384 / 4; i += kWbRailsThreads)
  {
    reinterpret_cast<uint4 *>(pack)[i] = reinterpret_cast<const uint4 *>(B.wbpack)[i];
  }
  __syncthreads();
  const size_t n32 = (size_t)B.n * 32;
  const uint32_t lo32 = B.lo * 32u, len32 = (B.len != 0u) ? B.len * 32u : (uint32_t)n32;   // the slice of every channel's row
  const uint32_t sq = len32 / 4;                         // quads per channel in the slice
  // work items of kWbRailsThreads quads, none across a channel boundary: one 32-bit division per item, none per quad
  // (the flat 64-bit index of the first version cost three 64-bit divisions per quad: more than the lookups)
  const uint32_t per_ch = (sq + kWbRailsThreads - 1) / kWbRailsThreads;
  const uint32_t items = per_ch * B.n_channels;
  for (uint32_t it = blockIdx.x; it < items; it += gridDim.x)
  {
    const uint32_t c = it / per_ch;
    const uint32_t q = (it - c * per_ch) * kWbRailsThreads + threadIdx.x;   // quad of the channel's slice
    if (q >= sq)
    {
      continue;
    }
    uint32_t *cell = B.wb + (size_t)c * n32 + lo32 + 4 * (size_t)q;
    const uint4 ph = *reinterpret_cast<const uint4 *>(cell);
    const uint32_t pb[4] = {ph.x, ph.y, ph.z, ph.w};
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
      w[j] = wb_lookup(pack, pb[j]);
    }
    *reinterpret_cast<uint4 *>(cell) = make_uint4(w[0], w[1], w[2], w[3]);
    if ((size_t)lo32 + 4 * (size_t)q + 4 == n32)
    {
      B.wbtail_out[(size_t)c * 2] = w[2];
      B.wbtail_out[(size_t)c * 2 + 1] = w[3];
    }
  }
}

// ---- round 6: k_wb_tail = k_wb_rails + k_mod<WB_TAIL> in one pass, the rails never exist in memory ----------------
// WbFmModulator::modulateSignal does Nco::runFast, the x900 scaling and the interpolator per sample with nothing stored
// between (WbFmModulator.cc:583-637).  Rounds 2-5 wrote the rail pair of every 256 kS/s cell back over its phase
// (k_wb_rails) and read it again in the x8 cascade (k_mod<WB_TAIL>): 8 of the call's 24 bytes of traffic per cell beside
// the 16 bytes of output.  Here a wave reads a run of PHASES (4 bytes per cell, coalesced dwords), looks the rail pairs
// up in the 64 KiB table -- in LDS once per PERSISTENT workgroup, not once per tile --, takes the pair of the cell in
// front from its left neighbour lane (DPP: lane 0 from the last lane of the group in front) and runs stages 6-8 of the
// cascade in registers (tail_eight, the code k_mod's tail runs): 16 bytes out per cell, one coalesced store per lane.
// No LDS traffic but the table lookups, no barrier behind the table's load, waves independent of each other.
// Work item = (channel, run of 2048 cells = 64 input samples: k_mod's tile), eight rounds of 4 x 64 cells; items are
// dealt like k_mod's tiles: XCD x takes the channels x, x + 8, ... and its waves walk consecutive items (one contiguous
// output stream per L2).
struct WbTailParams
{
  const uint32_t *cells;    // [C][32 n] the Nco phase of every 256 kS/s sample (float bits; k_phase_rows' output)
  int8_t *out;              // [C][512 n]
  const uint32_t *wbpack;   // [16384] the rail-pair table (k_wb_rails)
  const uint32_t *wbtail;   // [C][2] the previous call's last two rail pairs ([1] is the one in front of cell 0)
  uint32_t *wbtail_out;     // [C][2] this call's
  uint32_t n, n_channels;   // input samples per channel
  uint32_t lo, len;         // the slice [lo, lo + len) of every channel in input samples (len 0: all)
};
#ifndef HRFD_WT_THREADS
#define HRFD_WT_THREADS 1024
#endif
constexpr int kWtThreads = HRFD_WT_THREADS;             // (1024: one table for sixteen waves, two workgroups per CU -- A/B in profiles/r6_*)
constexpr uint32_t kWtRun = 2048u;                        // cells per work item (64 input samples)
__global__ __launch_bounds__(kWtThreads) void k_wb_tail(const WbTailParams P)
{
  __shared__ __attribute__((aligned(16))) uint32_t pack[16384];
  for (int i = threadIdx.x; i < 16384 / 4; i += kWtThreads)
  {
    reinterpret_cast<uint4 *>(pack)[i] = reinterpret_cast<const uint4 *>(P.wbpack)[i];
  }
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t n32 = P.n * 32u;
  const uint32_t lo32 = P.lo * 32u;
  const uint32_t end32 = (P.len != 0u) ? min(n32, (P.lo + P.len) * 32u) : n32;
  const uint32_t runs = (end32 - lo32 + kWtRun - 1u) / kWtRun;          // work items per channel
  const uint32_t xcd = blockIdx.x & 7u, wi = blockIdx.x >> 3, wgs_x = gridDim.x >> 3;
  const uint32_t ch_x = (P.n_channels > xcd) ? (P.n_channels - xcd + 7u) / 8u : 0u;
  const uint32_t items_x = ch_x * runs;
  const int k8000 = tail_k8000();
  const uint32_t voff = lane * 16u;
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  for (uint32_t it = wi * (kWtThreads / 64) + wave; it < items_x; it += wgs_x * (kWtThreads / 64))
  {
    const uint32_t ci = it / runs;
    const uint32_t c = 8u * ci + xcd;
    const uint32_t q0 = lo32 + (it - ci * runs) * kWtRun;
    const uint32_t q1 = min(q0 + kWtRun, end32);
    const uint32_t *row = P.cells + (size_t)c * n32;
    int8_t *orow = P.out + (size_t)c * n32 * 16u;
    // the rail pair in front of the run, in every lane: the previous call's last pair, or the lookup of the cell in front
    uint32_t carry = (q0 == 0u) ? P.wbtail[(size_t)c * 2 + 1] : wb_lookup(pack, row[q0 - 1u]);
    uint32_t ph[4];
#pragma unroll
    for (int m = 0; m < 4; m++)
    {
      const uint32_t cell = q0 + 64u * m + lane;
      ph[m] = (cell < q1) ? row[cell] : 0u;
    }
#pragma nounroll
    for (uint32_t q = q0; q < q1; q += 256u)
    {
      // (the stores below are opaque to the compiler, which moves no load across them: the next round's phases first)
      uint32_t nph[4];
#pragma unroll
      for (int m = 0; m < 4; m++)
      {
        const uint32_t cell = q + 256u + 64u * m + lane;
        nph[m] = (cell < q1) ? row[cell] : 0u;
      }
      uint32_t rails[4];
#pragma unroll
      for (int m = 0; m < 4; m++)
      {
        rails[m] = wb_lookup(pack, ph[m]);
      }
      if (q + 256u >= n32)
      {
        // the call's last two pairs are the next call's history (n32 is a multiple of 32: both in one group of 64)
#pragma unroll
        for (int m = 0; m < 4; m++)
        {
          const uint32_t cell = q + 64u * m + lane;
          if (cell + 2u == n32 || cell + 1u == n32)
          {
            P.wbtail_out[(size_t)c * 2 + (cell + 2u - n32)] = rails[m];
          }
        }
      }
#pragma unroll
      for (int m = 0; m < 4; m++)
      {
        const uint32_t left = (m == 0) ? carry : rails[m - 1];
        const uint32_t pv = shr1(rails[m], ror1(left));    // the pair of the cell in front: lane - 1's, lane 0: the last lane's of the group in front
        const int v[2][2] = {{(int)(int16_t)(rails[m] & 0xffffu), (int)(int16_t)(pv & 0xffffu)}, {(int)rails[m] >> 16, (int)pv >> 16}};
        const uint4 w4 = tail_eight(v, k8000);
        const uint32_t g = q + 64u * m;                      // the group's first cell (uniform)
        if (g + lane < q1)
        {
          int8_t *base = orow + (size_t)g * 16u;
          const u4 d4 = u4{w4.x, w4.y, w4.z, w4.w};
          // (scalar base, the lane's offset one register for the whole kernel; s_nop 0: see k_mod's tail)
          asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 0" : : "v"(voff), "v"(d4), "s"(base));
        }
      }
      carry = rails[3];
#pragma unroll
      for (int m = 0; m < 4; m++)
      {
        ph[m] = nph[m];
      }
    }
  }
}

// =============================================================================
//  Nco (Nco/Nco.cc, Nco/PhaseAccumulator.cc): one oscillator per thread
// =============================================================================
struct NcoParams
{
  float *acc;               // [C] phase accumulator
  const float *step;        // [C] (float)((2*M_PI*f)/fs)
  const float *sin_t;       // [16384] host-built (sinf of an accumulated float angle)
  const float *cos_t;
  float *i_out, *q_out;     // [C][count]
  uint32_t n_channels, count;
  int fast;
  int libm_fma;             // which build of glibc's sinf / cosf the host has (glibc_sinf)
};

__global__ void k_nco(const NcoParams N)
{
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N.n_channels)
  {
    return;
  }
  const double pi = 3.14159265358979323846, two_pi = 6.283185307179586476925286766559;
  float acc = N.acc[c];
  const float step = N.step[c];
  for (uint32_t k = 0; k < N.count; k++)
  {
    // PhaseAccumulator::run (:157-181): return the current phase, then advance
    // and wrap with double compares / double subtraction stored to float
    const float phase = acc;
    acc = acc + step;
    while ((double)acc > pi)
    {
      acc = (float)((double)acc - two_pi);
    }
    while ((double)acc < -pi)
    {
      acc = (float)((double)acc + two_pi);
    }
    float iv, qv;
    if (N.fast)
    {
      // Nco::runFast (:222-257): (int16_t)(phase * 16384 / (2*M_PI)) + 8192, clamped
      const float scaled = phase * 16384.0f;
      int idx = (int)(short)(int)((double)scaled / two_pi);
      idx += 8192;
      idx = max(0, min(16383, idx));
      iv = N.cos_t[idx];
      qv = N.sin_t[idx];
    }
    else
    {
      // Nco::run (:186-199) calls libm cosf/sinf: glibc's algorithm, restated (glibc_cosf: bit for bit)
      glibc_sincosf(phase, N.libm_fma, qv, iv);
    }
    N.i_out[(size_t)c * N.count + k] = iv;
    N.q_out[(size_t)c * N.count + k] = qv;
  }
  N.acc[c] = acc;
}

} // namespace hrfd
