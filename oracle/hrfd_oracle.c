/* oracle/hrfd_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference demodulation / modulation hot path
 * (wizardyesterday/HackRfDiags).  See hrfd_oracle.h for who may use it.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit against
 * the compiled reference (oracle/_ref/libhrfd_ref.so, built from the reference's
 * own sources by oracle/Makefile) in tests/test_oracle_vs_ref.py, and against
 * the golden vectors committed under tests/golden/ (generated from that same
 * reference build by tests/golden/make_golden.py).  The reference's own test
 * programs pin nothing on this path.
 *
 * Style: closed-form, block-at-a-time stage equations
 * with explicit carried state, NOT the reference's per-sample ring buffers.
 * Notation: a Q15 stage D(N,M,h) computes
 *     y[m] = (int16)((16384 + sum_{k<N} hq[k] * x[M*m + M-1-k]) >> 15)
 * with hq[k] = (int16)roundf(h[k]*32768), int32 wrap-around accumulation and
 * low-16-bit narrowing (Decimator_int16.cc:55-63,176-249,321-362).
 *
 * Compile with -ffp-contract=off: the float steps must round after every
 * multiply and add, as the reference (g++ -O3, x86-64 SSE2) does.
 */
#define _GNU_SOURCE
#include "hrfd_oracle.h"
#include "hrfd_oracle_tables.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAXTAPS 128      /* the longest table of the path has 40 taps; the reference's own Decimator_int16 test program uses 80 */

/* ------------------------------------------------------------------ helpers */

/* (int16_t)f as x86-64 executes it: cvttss2si to int32 ("integer indefinite"
 * 0x80000000 for NaN / out of range), then keep the low 16 bits.
 * Call sites: WbFmDemodulator.cc:476, FmDemodulator.cc:567, AmDemodulator.cc:466,
 * SsbDemodulator.cc:593, SsbModulator.cc:682. */
static inline int16_t f2i16(float f)
{
  int32_t v;
  if (f >= -2147483648.0f && f < 2147483648.0f)
  {
    v = (int32_t)f;
  }
  else
  {
    v = INT32_MIN;
  }
  return (int16_t)(uint16_t)((uint32_t)v & 0xffffu);
}

/* Same for a double operand (Nco.cc:231: (int16_t)(phase*16384/(2*M_PI))). */
static inline int16_t d2i16(double d)
{
  int32_t v;
  if (d > -2147483649.0 && d < 2147483648.0)
  {
    v = (int32_t)d;
  }
  else
  {
    v = INT32_MIN;
  }
  return (int16_t)(uint16_t)((uint32_t)v & 0xffffu);
}

/* deltaTheta wrap, WbFmDemodulator.cc:417-425 / FmDemodulator.cc:509-517:
 * compares and subtractions are carried out in double, the variable is float. */
static inline float wrap_pi(float d)
{
  while (d > M_PI)
  {
    d = (float)((double)d - (2 * M_PI));
  }
  while (d < (-M_PI))
  {
    d = (float)((double)d + (2 * M_PI));
  }
  return d;
}

/* ------------------------------------------------------------------ Q15 stage
 * One struct serves Decimator_int16 (M>1), FirFilter_int16 (M=1). */
typedef struct
{
  int n;                      /* taps */
  int m;                      /* decimation factor */
  int16_t hq[ORC_MAXTAPS];    /* quantised taps */
  int16_t tail[ORC_MAXTAPS];  /* last n-1 inputs, tail[n-2] newest */
  uint32_t phase;             /* inputs consumed since reset, mod m */
} q15_t;

static void quantise(const float *h, int n, int16_t *hq)
{
  /* Decimator_int16.cc:55-63: float multiply, round() on a float argument
   * (roundf under the C++ overloads), then (int16_t) narrowing: 1.0 -> -32768. */
  for (int i = 0; i < n; i++)
  {
    float scaled = h[i] * 32768;
    scaled = roundf(scaled);
    hq[i] = f2i16(scaled);
  }
}

static void q15_reset(q15_t *s)
{
  memset(s->tail, 0, sizeof(s->tail));
  s->phase = 0;
}

static void taps_fit(int n)
{
  if (n < 1 || n > ORC_MAXTAPS)
  {
    fprintf(stderr, "hrfd oracle: %d taps (1..%d)\n", n, ORC_MAXTAPS);
    abort();
  }
}

static void q15_init(q15_t *s, const float *h, int n, int m)
{
  taps_fit(n);
  memset(s, 0, sizeof(*s));
  s->n = n;
  s->m = m;
  quantise(h, n, s->hq);
  q15_reset(s);
}

/* Consume cnt inputs, emit one output per completed group of m. */
static uint32_t q15_run(q15_t *s, const int16_t *x, uint32_t cnt, int16_t *y)
{
  const int n = s->n;
  const uint32_t m = (uint32_t)s->m;
  int16_t *ext = (int16_t *)malloc(((size_t)(n - 1) + cnt + 1) * sizeof(int16_t));
  uint32_t out = 0;

  memcpy(ext, s->tail, (size_t)(n - 1) * sizeof(int16_t));
  if (cnt > 0)
  {
    memcpy(ext + (n - 1), x, (size_t)cnt * sizeof(int16_t));
  }
  for (uint32_t i = (m - 1u - s->phase) % m; i < cnt; i += m)
  {
    const int16_t *p = ext + (n - 1) + i;      /* -> x[i] */
    uint32_t acc = 1u << 14;                   /* rounding constant, wraps mod 2^32 */
    for (int k = 0; k < n; k++)
    {
      acc += (uint32_t)((int32_t)s->hq[k] * (int32_t)p[-k]);
    }
    y[out++] = (int16_t)(uint16_t)(((uint32_t)((int32_t)acc >> 15)) & 0xffffu);
  }
  memcpy(s->tail, ext + cnt, (size_t)(n - 1) * sizeof(int16_t));
  s->phase = (s->phase + cnt) % m;
  free(ext);
  return out;
}

/* ------------------------------------------------------------------ Q15 polyphase x L
 * Interpolator_int16.cc:267-333 (coefficients), :398-418 (interpolate):
 *   y[L*n + i] = (int16)((16384 + sum_j hq[i + j*L] * x[n-j]) >> 15), j < N/L. */
typedef struct
{
  int n, l, q;                /* prototype taps, factor, taps per phase */
  int16_t hq[ORC_MAXTAPS];
  int16_t tail[ORC_MAXTAPS];  /* last q-1 inputs */
} q15i_t;

static void q15i_init(q15i_t *s, const float *h, int n, int l)
{
  taps_fit(n);
  memset(s, 0, sizeof(*s));
  s->n = n;
  s->l = l;
  s->q = n / l;
  quantise(h, n, s->hq);
}

static void q15i_reset(q15i_t *s)
{
  memset(s->tail, 0, sizeof(s->tail));
}

static void q15i_run(q15i_t *s, const int16_t *x, uint32_t cnt, int16_t *y)
{
  const int q = s->q, l = s->l;
  int16_t *ext = (int16_t *)malloc(((size_t)(q - 1) + cnt + 1) * sizeof(int16_t));
  memcpy(ext, s->tail, (size_t)(q - 1) * sizeof(int16_t));
  if (cnt > 0)
  {
    memcpy(ext + (q - 1), x, (size_t)cnt * sizeof(int16_t));
  }
  for (uint32_t i = 0; i < cnt; i++)
  {
    const int16_t *p = ext + (q - 1) + i;
    for (int ph = 0; ph < l; ph++)
    {
      uint32_t acc = 1u << 14;
      for (int j = 0; j < q; j++)
      {
        acc += (uint32_t)((int32_t)s->hq[ph + j * l] * (int32_t)p[-j]);
      }
      y[(size_t)i * l + ph] = (int16_t)(uint16_t)(((uint32_t)((int32_t)acc >> 15)) & 0xffffu);
    }
  }
  memcpy(s->tail, ext + cnt, (size_t)(q - 1) * sizeof(int16_t));
  free(ext);
}

/* ------------------------------------------------------------------ float FIR / IIR
 * FirFilter.cc:144-185: y = 0; for k: y = y + h[k]*x[n-k]  (that order).
 * IirFilter.cc:161-176: y = fir(x); y -= (0 + sum a[k]*ypast[k]); push y
 * (for the pairing of a[k] with past outputs see iirf_step). */
typedef struct
{
  int n;
  float h[ORC_MAXTAPS];
  float x[ORC_MAXTAPS];       /* x[0] = newest */
} firf_t;

static void firf_init(firf_t *s, const float *h, int n)
{
  taps_fit(n);
  memset(s, 0, sizeof(*s));
  s->n = n;
  memcpy(s->h, h, (size_t)n * sizeof(float));
}

static void firf_reset(firf_t *s)
{
  memset(s->x, 0, sizeof(s->x));
}

static inline float firf_step(firf_t *s, float x)
{
  float y = 0;
  memmove(&s->x[1], &s->x[0], (size_t)(s->n - 1) * sizeof(float));
  s->x[0] = x;
  for (int k = 0; k < s->n; k++)
  {
    y = y + (s->h[k] * s->x[k]);
  }
  return y;
}

typedef struct
{
  firf_t num;
  int na;
  float a[ORC_MAXTAPS];
  float y[ORC_MAXTAPS];       /* y[0] = newest */
} iirf_t;

static void iirf_init(iirf_t *s, const float *b, int nb, const float *a, int na)
{
  memset(s, 0, sizeof(*s));
  firf_init(&s->num, b, nb);
  taps_fit(na);
  s->na = na;
  memcpy(s->a, a, (size_t)na * sizeof(float));
}

static void iirf_reset(iirf_t *s)
{
  firf_reset(&s->num);
  memset(s->y, 0, sizeof(s->y));
}

static inline float iirf_step(iirf_t *s, float x)
{
  float y = firf_step(&s->num, x);
  float r = 0;
  /* IirFilter::filterRecursive (:199-229) starts at ringBufferIndex, which after
   * shiftSampleIn (:245-260) points at the OLDEST stored output, then walks
   * backwards: a[0] pairs with y[n-na], a[k>=1] with y[n-k].  For the one-tap
   * denominators on this path (de-emphasis, dc removal) that is just y[n-1]. */
  for (int k = 0; k < s->na; k++)
  {
    int idx = (k == 0) ? (s->na - 1) : (k - 1);   /* index into y[] (0 = newest) */
    r = r + (s->a[k] * s->y[idx]);
  }
  y -= r;
  memmove(&s->y[1], &s->y[0], (size_t)(s->na - 1) * sizeof(float));
  s->y[0] = y;
  return y;
}

/* ------------------------------------------------------------------ tables */
static float g_atan2[256][256];
static int32_t g_dbfs[257];
static int g_tables_ready = 0;

static void build_tables(void)
{
  if (g_tables_ready)
  {
    return;
  }
  /* WbFmDemodulator.cc:137-148 / FmDemodulator.cc:159-170: libm atan2 in double,
   * stored as float, indexed [q+128][i+128]. */
  for (int x = 0; x < 256; x++)
  {
    for (int y = 0; y < 256; y++)
    {
      double xa = (double)x - 128;
      double ya = (double)y - 128;
      g_atan2[y][x] = (float)atan2(ya, xa);
    }
  }
  /* DbfsCalculator.cc:58-65: 20*log10((float)i) resolves to log10f in C++. */
  for (int i = 1; i <= 256; i++)
  {
    float db = 20 * log10f((float)i);
    g_dbfs[i] = (int32_t)db;
  }
  g_dbfs[0] = g_dbfs[1];
  g_tables_ready = 1;
}

void orc_atan2_lut(float *out)
{
  build_tables();
  memcpy(out, g_atan2, sizeof(g_atan2));
}

void orc_dbfs_table(int32_t *out)
{
  build_tables();
  memcpy(out, g_dbfs, sizeof(g_dbfs));
}

/* ------------------------------------------------------------------ demodulators */
typedef struct
{
  float gain;
  float theta_prev;
  iirf_t deemph;
  q15_t d1, d2, d3;
  float *stream;              /* demodulatedData of the last call */
  uint32_t stream_len;
} wbfm_t;

static void wbfm_init(wbfm_t *s)
{
  memset(s, 0, sizeof(*s));
  s->gain = (float)(256000 / (2 * M_PI));                 /* WbFmDemodulator.cc:151 */
  iirf_init(&s->deemph, DEEMPH_B, 2, DEEMPH_A, 1);
  q15_init(&s->d1, WBFM_D1, 8, 4);
  q15_init(&s->d2, POST_D12, 12, 4);
  q15_init(&s->d3, AUDIO_D40, 40, 2);
  s->theta_prev = 0;
}

static void wbfm_reset(wbfm_t *s)
{
  /* WbFmDemodulator.cc:265-278 -- NB the de-emphasis IIR is NOT reset. */
  q15_reset(&s->d1);
  q15_reset(&s->d2);
  q15_reset(&s->d3);
  s->theta_prev = 0;
}

/* WbFmDemodulator::demodulateSignal (:381-439) + createPcmData (:460-500). */
static uint32_t wbfm_process(wbfm_t *s, const int8_t *iq, uint32_t bytes, int16_t *pcm)
{
  uint32_t count = bytes / 2;
  float k = s->gain / 75000;
  int16_t *s16 = (int16_t *)malloc(((size_t)count + 1) * sizeof(int16_t));
  int16_t *t1 = (int16_t *)malloc(((size_t)count / 4 + 2) * sizeof(int16_t));
  int16_t *t2 = (int16_t *)malloc(((size_t)count / 16 + 2) * sizeof(int16_t));
  uint32_t n1, n2, n3;

  k *= 32767;
  s->stream = (float *)realloc(s->stream, ((size_t)count + 1) * sizeof(float));
  s->stream_len = count;
  for (uint32_t i = 0; i < count; i++)
  {
    uint8_t ii = (uint8_t)((uint8_t)iq[2 * i] + 128);
    uint8_t qi = (uint8_t)((uint8_t)iq[2 * i + 1] + 128);
    float theta = g_atan2[qi][ii];
    float d = theta - s->theta_prev;
    d = wrap_pi(d);
    s->stream[i] = iirf_step(&s->deemph, k * d);
    s->theta_prev = theta;
    s16[i] = f2i16(s->stream[i]);
  }
  n1 = q15_run(&s->d1, s16, count, t1);
  n2 = q15_run(&s->d2, t1, n1, t2);
  n3 = q15_run(&s->d3, t2, n2, pcm);
  free(s16);
  free(t1);
  free(t2);
  return n3;
}

typedef struct
{
  float gain;
  q15_t ti, tq;               /* tuner decimators */
  firf_t diff;
  q15_t d2, d3;
} fm_t;

static void fm_init(fm_t *s)
{
  memset(s, 0, sizeof(*s));
  s->gain = (float)(64000 / (2 * M_PI));                  /* FmDemodulator.cc:173 */
  q15_init(&s->ti, FM_TUNER_D32, 32, 4);
  q15_init(&s->tq, FM_TUNER_D32, 32, 4);
  firf_init(&s->diff, FM_DIFF, 7);
  q15_init(&s->d2, POST_D12, 12, 4);
  q15_init(&s->d3, AUDIO_D40, 40, 2);
}

static void fm_reset(fm_t *s)
{
  q15_reset(&s->ti);
  q15_reset(&s->tq);
  q15_reset(&s->d2);
  q15_reset(&s->d3);
  firf_reset(&s->diff);
}

/* FmDemodulator::reduceSampleRate (:395-442), demodulateSignal (:479-529),
 * createPcmData (:551-585). */
static uint32_t fm_process(fm_t *s, const int8_t *iq, uint32_t bytes, int16_t *pcm)
{
  uint32_t count = bytes / 2;
  int16_t *xi = (int16_t *)malloc(((size_t)count + 1) * sizeof(int16_t));
  int16_t *xq = (int16_t *)malloc(((size_t)count + 1) * sizeof(int16_t));
  int16_t *di = (int16_t *)malloc(((size_t)count / 4 + 2) * sizeof(int16_t));
  int16_t *dq = (int16_t *)malloc(((size_t)count / 4 + 2) * sizeof(int16_t));
  int16_t *t2 = (int16_t *)malloc(((size_t)count / 16 + 2) * sizeof(int16_t));
  float k = s->gain / 15000;
  uint32_t n, n2, n3;

  k *= 32767;
  for (uint32_t i = 0; i < count; i++)
  {
    xi[i] = (int16_t)iq[2 * i];
    xq[i] = (int16_t)iq[2 * i + 1];
  }
  n = q15_run(&s->ti, xi, count, di);
  (void)q15_run(&s->tq, xq, count, dq);
  for (uint32_t i = 0; i < n; i++)
  {
    /* low byte of the int16 sample, biased by 128 (FmDemodulator.cc:495-496) */
    uint8_t ii = (uint8_t)((uint8_t)di[i] + 128);
    uint8_t qi = (uint8_t)((uint8_t)dq[i] + 128);
    float theta = g_atan2[qi][ii];
    float d = firf_step(&s->diff, theta);
    d = wrap_pi(d);
    di[i] = f2i16(k * d);                      /* reuse di as the int16 stream */
  }
  n2 = q15_run(&s->d2, di, n, t2);
  n3 = q15_run(&s->d3, t2, n2, pcm);
  free(xi); free(xq); free(di); free(dq); free(t2);
  return n3;
}

typedef struct
{
  float gain;
  int lsb;
  q15_t s1[2], s2[2], s3[2];  /* [0]=I rail, [1]=Q rail */
  q15_t delay, hilbert;       /* SSB only */
  iirf_t dcrem;
} amssb_t;

static void amssb_init(amssb_t *s)
{
  memset(s, 0, sizeof(*s));
  s->gain = 300;                                          /* AmDemodulator.cc:102 */
  s->lsb = 1;
  for (int r = 0; r < 2; r++)
  {
    q15_init(&s->s1[r], AM_D1, 8, 4);
    q15_init(&s->s2[r], AM_D2, 12, 4);
    q15_init(&s->s3[r], AM_D3, 16, 2);
  }
  q15_init(&s->delay, SSB_DELAY, 16, 1);
  q15_init(&s->hilbert, SSB_HILBERT, 31, 1);
  iirf_init(&s->dcrem, DCREM_B, 2, DCREM_A, 1);
}

static void amssb_reset(amssb_t *s)
{
  for (int r = 0; r < 2; r++)
  {
    q15_reset(&s->s1[r]);
    q15_reset(&s->s2[r]);
    q15_reset(&s->s3[r]);
  }
  q15_reset(&s->delay);
  q15_reset(&s->hilbert);
  iirf_reset(&s->dcrem);
}

/* {Am,Ssb}Demodulator::reduceSampleRate (AmDemodulator.cc:339-408,
 * SsbDemodulator.cc:462-529): /4 /4 /2 on each rail, int16 kept. */
static uint32_t amssb_reduce(amssb_t *s, const int8_t *iq, uint32_t bytes,
                             int16_t *oi, int16_t *oq)
{
  uint32_t count = bytes / 2, n = 0;
  int16_t *x = (int16_t *)malloc(((size_t)count + 1) * sizeof(int16_t));
  int16_t *t1 = (int16_t *)malloc(((size_t)count / 4 + 2) * sizeof(int16_t));
  int16_t *t2 = (int16_t *)malloc(((size_t)count / 16 + 2) * sizeof(int16_t));
  for (int r = 0; r < 2; r++)
  {
    uint32_t n1, n2;
    for (uint32_t i = 0; i < count; i++)
    {
      x[i] = (int16_t)iq[2 * i + r];
    }
    n1 = q15_run(&s->s1[r], x, count, t1);
    n2 = q15_run(&s->s2[r], t1, n1, t2);
    n = q15_run(&s->s3[r], t2, n2, r == 0 ? oi : oq);
  }
  free(x); free(t1); free(t2);
  return n;
}

/* AmDemodulator::demodulateSignal (:434-471). */
static uint32_t am_process(amssb_t *s, const int8_t *iq, uint32_t bytes, int16_t *pcm)
{
  int16_t *di = (int16_t *)malloc(((size_t)bytes / 64 + 2) * sizeof(int16_t));
  int16_t *dq = (int16_t *)malloc(((size_t)bytes / 64 + 2) * sizeof(int16_t));
  uint32_t n = amssb_reduce(s, iq, bytes, di, dq);
  for (uint32_t i = 0; i < n; i++)
  {
    int16_t im = (int16_t)abs((int)di[i]);
    int16_t qm = (int16_t)abs((int)dq[i]);
    int16_t mag;
    float out;
    if (im > qm)
    {
      mag = (int16_t)(im + (qm >> 1));
    }
    else
    {
      mag = (int16_t)(qm + (im >> 1));
    }
    out = iirf_step(&s->dcrem, (float)mag);
    pcm[i] = f2i16(s->gain * out);
  }
  free(di); free(dq);
  return n;
}

/* SsbDemodulator::demodulateSignal (:563-598). */
static uint32_t ssb_process(amssb_t *s, const int8_t *iq, uint32_t bytes, int16_t *pcm)
{
  int16_t *di = (int16_t *)malloc(((size_t)bytes / 64 + 2) * sizeof(int16_t));
  int16_t *dq = (int16_t *)malloc(((size_t)bytes / 64 + 2) * sizeof(int16_t));
  int16_t *id = (int16_t *)malloc(((size_t)bytes / 64 + 2) * sizeof(int16_t));
  int16_t *qh = (int16_t *)malloc(((size_t)bytes / 64 + 2) * sizeof(int16_t));
  uint32_t n = amssb_reduce(s, iq, bytes, di, dq);
  (void)q15_run(&s->delay, di, n, id);
  (void)q15_run(&s->hilbert, dq, n, qh);
  for (uint32_t i = 0; i < n; i++)
  {
    float v;
    if (s->lsb)
    {
      v = (float)((int)id[i] - (int)qh[i]);
    }
    else
    {
      v = (float)((int)id[i] + (int)qh[i]);
    }
    v = iirf_step(&s->dcrem, v);
    pcm[i] = f2i16(s->gain * v);
  }
  free(di); free(dq); free(id); free(qh);
  return n;
}

/* ------------------------------------------------------------------ inner API */
struct orc_demod
{
  int mode;
  wbfm_t wbfm;
  fm_t fm;
  amssb_t amssb;
};

static void demod_init(struct orc_demod *h, int mode)
{
  build_tables();
  memset(h, 0, sizeof(*h));
  h->mode = mode;
  wbfm_init(&h->wbfm);
  fm_init(&h->fm);
  amssb_init(&h->amssb);
  h->amssb.lsb = (mode != ORC_USB);
}

static uint32_t demod_process(struct orc_demod *h, const int8_t *iq, uint32_t bytes, int16_t *pcm)
{
  switch (h->mode)
  {
    case ORC_AM: return am_process(&h->amssb, iq, bytes, pcm);
    case ORC_FM: return fm_process(&h->fm, iq, bytes, pcm);
    case ORC_WBFM: return wbfm_process(&h->wbfm, iq, bytes, pcm);
    case ORC_LSB:
    case ORC_USB: return ssb_process(&h->amssb, iq, bytes, pcm);
    default: return 0;
  }
}

orc_demod *orc_demod_create(int mode)
{
  struct orc_demod *h = (struct orc_demod *)malloc(sizeof(*h));
  demod_init(h, mode);
  return h;
}

void orc_demod_destroy(orc_demod *h)
{
  free(h->wbfm.stream);
  free(h);
}

void orc_demod_reset(orc_demod *h)
{
  switch (h->mode)
  {
    case ORC_AM:
    case ORC_LSB:
    case ORC_USB: amssb_reset(&h->amssb); break;
    case ORC_FM: fm_reset(&h->fm); break;
    case ORC_WBFM: wbfm_reset(&h->wbfm); break;
    default: break;
  }
}

void orc_demod_set_gain(orc_demod *h, float gain)
{
  h->wbfm.gain = gain;
  h->fm.gain = gain;
  h->amssb.gain = gain;
}

void orc_demod_set_sideband(orc_demod *h, int lsb)
{
  h->amssb.lsb = lsb ? 1 : 0;
}

uint32_t orc_demod_process(orc_demod *h, const int8_t *iq256, uint32_t bytes,
                           int16_t *pcm, uint32_t pcm_cap)
{
  int16_t *tmp = (int16_t *)malloc(((size_t)bytes / 2 + 8) * sizeof(int16_t));
  uint32_t n = demod_process(h, iq256, bytes, tmp);
  if (n > pcm_cap)
  {
    n = pcm_cap;
  }
  memcpy(pcm, tmp, (size_t)n * sizeof(int16_t));
  free(tmp);
  return n;
}

/* ------------------------------------------------------------------ outer API */
struct orc_rx
{
  q15_t fe[2][3];             /* [rail][stage] half-band decimators */
  int mode;
  int32_t threshold;
  int tracking;               /* SignalTracker state */
  uint32_t magnitude;
  /* one instance of each demodulator, like Radio.cc:179-197 */
  struct orc_demod am, fm, wbfm, ssb;
};

orc_rx *orc_rx_create(void)
{
  struct orc_rx *h = (struct orc_rx *)malloc(sizeof(*h));
  build_tables();
  memset(h, 0, sizeof(*h));
  for (int r = 0; r < 2; r++)
  {
    q15_init(&h->fe[r][0], HB1, 3, 2);
    q15_init(&h->fe[r][1], HB2, 3, 2);
    q15_init(&h->fe[r][2], HB3, 3, 2);
  }
  h->mode = ORC_NONE;
  h->threshold = -200;                                    /* IqDataProcessor.cc:121 */
  h->tracking = 0;
  demod_init(&h->am, ORC_AM);
  demod_init(&h->fm, ORC_FM);
  demod_init(&h->wbfm, ORC_WBFM);
  demod_init(&h->ssb, ORC_LSB);
  return h;
}

void orc_rx_destroy(orc_rx *h)
{
  free(h->am.wbfm.stream);
  free(h->fm.wbfm.stream);
  free(h->wbfm.wbfm.stream);
  free(h->ssb.wbfm.stream);
  free(h);
}

void orc_rx_set_mode(orc_rx *h, int mode)
{
  /* IqDataProcessor::setDemodulatorMode (:346-375) */
  h->mode = mode;
  if (mode == ORC_LSB)
  {
    h->ssb.amssb.lsb = 1;
    h->ssb.mode = ORC_LSB;
  }
  else if (mode == ORC_USB)
  {
    h->ssb.amssb.lsb = 0;
    h->ssb.mode = ORC_USB;
  }
}

void orc_rx_set_gain(orc_rx *h, int mode, float gain)
{
  switch (mode)
  {
    case ORC_AM: orc_demod_set_gain(&h->am, gain); break;
    case ORC_FM: orc_demod_set_gain(&h->fm, gain); break;
    case ORC_WBFM: orc_demod_set_gain(&h->wbfm, gain); break;
    case ORC_LSB:
    case ORC_USB: orc_demod_set_gain(&h->ssb, gain); break;
    default: break;
  }
}

void orc_rx_set_threshold(orc_rx *h, int32_t threshold)
{
  h->threshold = threshold;
}

uint32_t orc_rx_wbfm_float_stream(orc_rx *h, float *out, uint32_t cap)
{
  uint32_t n = h->wbfm.wbfm.stream_len;
  if (n > cap)
  {
    n = cap;
  }
  if (n > 0)
  {
    memcpy(out, h->wbfm.wbfm.stream, (size_t)n * sizeof(float));
  }
  return n;
}

/* DbfsCalculator::convertMagnitudeToDbFs (:111-147), word length 7 bits
 * (SignalDetector.cc ctor): full scale 127, (uint32)(20*log10(127.0)) = 42. */
static int32_t magnitude_to_dbfs(uint32_t magnitude)
{
  const uint32_t full_scale = 127;
  const uint32_t full_scale_db = (uint32_t)(20 * log10((double)full_scale));
  int32_t decibels = 0;
  int32_t v;
  if (magnitude > full_scale)
  {
    magnitude = full_scale;
  }
  while (magnitude > 256)
  {
    magnitude /= 2;
    decibels += 6;
  }
  v = g_dbfs[magnitude];
  v += decibels;
  v = (int32_t)((uint32_t)v - full_scale_db);
  return v;
}

uint32_t orc_rx_process(orc_rx *h, const int8_t *iq, uint32_t bytes, uint32_t gain_db,
                        int16_t *pcm, uint32_t pcm_cap, uint32_t *magnitude,
                        int *signal_allowed, int8_t *iq256_out)
{
  uint32_t n_in = bytes / 2;
  uint32_t n3 = 0, dec_bytes, n_pcm = 0;
  int16_t *x = (int16_t *)malloc(((size_t)n_in + 1) * sizeof(int16_t));
  int16_t *t1 = (int16_t *)malloc(((size_t)n_in / 2 + 2) * sizeof(int16_t));
  int16_t *t2 = (int16_t *)malloc(((size_t)n_in / 4 + 2) * sizeof(int16_t));
  int16_t *t3 = (int16_t *)malloc(((size_t)n_in / 8 + 2) * sizeof(int16_t));
  int8_t *dec = (int8_t *)calloc((size_t)n_in / 4 + 16, 1);
  int present, allowed;
  uint32_t mag_sum = 0, mag_n;
  int32_t dbfs;

  /* A2: IqDataProcessor::reduceSampleRate (:429-500): three half-band /2 stages
   * per rail, (int8_t) narrowing, re-interleave. */
  for (int r = 0; r < 2; r++)
  {
    uint32_t n1, n2;
    for (uint32_t i = 0; i < n_in; i++)
    {
      x[i] = (int16_t)iq[2 * i + r];
    }
    n1 = q15_run(&h->fe[r][0], x, n_in, t1);
    n2 = q15_run(&h->fe[r][1], t1, n1, t2);
    n3 = q15_run(&h->fe[r][2], t2, n2, t3);
    for (uint32_t j = 0; j < n3; j++)
    {
      dec[2 * j + r] = (int8_t)(uint8_t)((uint16_t)t3[j] & 0xffu);
    }
  }
  dec_bytes = 2 * n3;

  /* A4: upconvertByFsOver4 (:771-815), period-4 rotation by index within the call. */
  for (uint32_t i = 0; i < dec_bytes; i += 8)
  {
    int8_t a, b;
    a = dec[i + 2]; b = dec[i + 3];
    dec[i + 2] = (int8_t)(uint8_t)(0u - (uint8_t)b); dec[i + 3] = a;
    a = dec[i + 4]; b = dec[i + 5];
    dec[i + 4] = (int8_t)(uint8_t)(0u - (uint8_t)a); dec[i + 5] = (int8_t)(uint8_t)(0u - (uint8_t)b);
    a = dec[i + 6]; b = dec[i + 7];
    dec[i + 6] = b; dec[i + 7] = (int8_t)(uint8_t)(0u - (uint8_t)a);
  }

  /* A5: SignalDetector::detectSignal (:205-274) */
  mag_n = dec_bytes / 2;
  for (uint32_t j = 0; j < mag_n; j++)
  {
    uint8_t im = (uint8_t)abs((int)dec[2 * j]);
    uint8_t qm = (uint8_t)abs((int)dec[2 * j + 1]);
    uint8_t m;
    if (im > qm)
    {
      m = (uint8_t)(im + (qm >> 1));
    }
    else
    {
      m = (uint8_t)(qm + (im >> 1));
    }
    mag_sum += m;
  }
  h->magnitude = (mag_n > 0) ? (mag_sum / mag_n) : 0;
  dbfs = magnitude_to_dbfs(h->magnitude);
  dbfs = (int32_t)((uint32_t)dbfs - gain_db);
  present = (dbfs >= h->threshold);

  /* SignalTracker::run (:104-146) + Squelch::run (:227-273): one tail block. */
  allowed = present || h->tracking;
  h->tracking = present;

  if (magnitude != NULL)
  {
    *magnitude = h->magnitude;
  }
  if (signal_allowed != NULL)
  {
    *signal_allowed = allowed;
  }
  if (iq256_out != NULL)
  {
    memcpy(iq256_out, dec, dec_bytes);
  }

  /* A6: mode dispatch (:991-1034). */
  if (allowed)
  {
    int16_t *tmp = (int16_t *)malloc(((size_t)dec_bytes / 2 + 8) * sizeof(int16_t));
    struct orc_demod *d = NULL;
    switch (h->mode)
    {
      case ORC_AM: d = &h->am; break;
      case ORC_FM: d = &h->fm; break;
      case ORC_WBFM: d = &h->wbfm; break;
      case ORC_LSB:
      case ORC_USB: d = &h->ssb; break;
      default: break;
    }
    if (d != NULL)
    {
      n_pcm = demod_process(d, dec, dec_bytes, tmp);
      if (n_pcm > pcm_cap)
      {
        n_pcm = pcm_cap;
      }
      if (pcm != NULL)
      {
        memcpy(pcm, tmp, (size_t)n_pcm * sizeof(int16_t));
      }
    }
    free(tmp);
  }
  free(x); free(t1); free(t2); free(t3); free(dec);
  return n_pcm;
}

/* ------------------------------------------------------------------ transmit */
struct orc_ssbmod
{
  int lsb;
  q15_t delay, hilbert;
  q15i_t ip[2][8];            /* [rail][stage] */
};

static void cascade_init(q15i_t ip[2][8], const float *stage1)
{
  for (int r = 0; r < 2; r++)
  {
    q15i_init(&ip[r][0], stage1, 40, 2);
    q15i_init(&ip[r][1], INTERP_HB8, 8, 2);
    q15i_init(&ip[r][2], INTERP_HB3, 4, 2);
    q15i_init(&ip[r][3], INTERP_HB8, 8, 2);
    q15i_init(&ip[r][4], INTERP_HB8, 8, 2);
    q15i_init(&ip[r][5], INTERP_HB3, 4, 2);
    q15i_init(&ip[r][6], INTERP_HB2, 4, 2);
    q15i_init(&ip[r][7], INTERP_HB1, 4, 2);
  }
}

/* eight x2 stages on one rail, then (int8_t) narrowing into the interleaved
 * output (SsbModulator.cc:499-619, interpolateSignal.cc:262-372). */
static void cascade_run(q15i_t ip[8], const int16_t *x, uint32_t n, int8_t *out, int rail)
{
  int16_t *a = (int16_t *)malloc(((size_t)n * 256 + 2) * sizeof(int16_t));
  int16_t *b = (int16_t *)malloc(((size_t)n * 256 + 2) * sizeof(int16_t));
  uint32_t cnt = n;
  memcpy(a, x, (size_t)n * sizeof(int16_t));
  for (int s = 0; s < 8; s++)
  {
    q15i_run(&ip[s], a, cnt, b);
    cnt *= 2;
    int16_t *t = a; a = b; b = t;
  }
  for (uint32_t i = 0; i < cnt; i++)
  {
    out[2 * (size_t)i + rail] = (int8_t)(uint8_t)((uint16_t)a[i] & 0xffu);
  }
  free(a); free(b);
}

orc_ssbmod *orc_ssbmod_create(int lsb)
{
  struct orc_ssbmod *h = (struct orc_ssbmod *)malloc(sizeof(*h));
  memset(h, 0, sizeof(*h));
  h->lsb = lsb ? 1 : 0;
  q15_init(&h->delay, SSB_DELAY, 16, 1);
  q15_init(&h->hilbert, SSB_HILBERT, 31, 1);
  cascade_init(h->ip, AUDIO_D40);
  return h;
}

void orc_ssbmod_destroy(orc_ssbmod *h)
{
  free(h);
}

void orc_ssbmod_reset(orc_ssbmod *h)
{
  q15_reset(&h->delay);
  q15_reset(&h->hilbert);
  for (int r = 0; r < 2; r++)
  {
    for (int s = 0; s < 8; s++)
    {
      q15i_reset(&h->ip[r][s]);
    }
  }
}

void orc_ssbmod_set_sideband(orc_ssbmod *h, int lsb)
{
  h->lsb = lsb ? 1 : 0;
}

uint32_t orc_ssbmod_process(orc_ssbmod *h, const int16_t *pcm, uint32_t n, int8_t *iq_out)
{
  int16_t *s = (int16_t *)calloc((size_t)n + 1, sizeof(int16_t));
  int16_t *id = (int16_t *)calloc((size_t)n + 1, sizeof(int16_t));
  int16_t *qh = (int16_t *)calloc((size_t)n + 1, sizeof(int16_t));
  /* SsbModulator::modulateSignal (:667-707) */
  for (uint32_t i = 0; i < n; i++)
  {
    float v = (float)pcm[i];
    v /= 2;
    s[i] = f2i16(v);
  }
  (void)q15_run(&h->delay, s, n, id);
  (void)q15_run(&h->hilbert, s, n, qh);
  if (!h->lsb)
  {
    for (uint32_t i = 0; i < n; i++)
    {
      qh[i] = (int16_t)(uint16_t)((0u - (uint16_t)qh[i]) & 0xffffu);
    }
  }
  cascade_run(h->ip[0], id, n, iq_out, 0);
  cascade_run(h->ip[1], qh, n, iq_out, 1);
  free(s); free(id); free(qh);
  return n << 9;                                           /* bytes, :512 */
}

struct orc_interp
{
  q15i_t ip[2][8];
};

orc_interp *orc_interp_create(void)
{
  struct orc_interp *h = (struct orc_interp *)malloc(sizeof(*h));
  memset(h, 0, sizeof(*h));
  cascade_init(h->ip, INTERPSIG_S1);
  return h;
}

void orc_interp_destroy(orc_interp *h)
{
  free(h);
}

uint32_t orc_interp_process(orc_interp *h, const int16_t *iq, uint32_t n_pairs, int8_t *iq_out)
{
  int16_t *xi = (int16_t *)malloc(((size_t)n_pairs + 1) * sizeof(int16_t));
  int16_t *xq = (int16_t *)malloc(((size_t)n_pairs + 1) * sizeof(int16_t));
  for (uint32_t i = 0; i < n_pairs; i++)
  {
    xi[i] = iq[2 * i];
    xq[i] = iq[2 * i + 1];
  }
  cascade_run(h->ip[0], xi, n_pairs, iq_out, 0);
  cascade_run(h->ip[1], xq, n_pairs, iq_out, 1);
  free(xi); free(xq);
  return n_pairs * 512u;
}

/* AmModulator (AmModulator.cc): modulateSignal :574-612, increaseSampleRate :406-483 (the same
 * eight x2 stages and tables as the SSB modulator), modulationIndex 0.8 by default (:218),
 * setModulationIndex accepts [0, 1] (:329-336). */
struct orc_ammod
{
  float index;
  q15i_t ip[2][8];
};

orc_ammod *orc_ammod_create(void)
{
  struct orc_ammod *h = (struct orc_ammod *)malloc(sizeof(*h));
  memset(h, 0, sizeof(*h));
  h->index = 0.8;
  cascade_init(h->ip, AUDIO_D40);
  return h;
}

void orc_ammod_destroy(orc_ammod *h)
{
  free(h);
}

void orc_ammod_reset(orc_ammod *h)
{
  for (int r = 0; r < 2; r++)
  {
    for (int s = 0; s < 8; s++)
    {
      q15i_reset(&h->ip[r][s]);
    }
  }
}

void orc_ammod_set_index(orc_ammod *h, float index)
{
  if ((index >= 0) && (index <= 1))
  {
    h->index = index;
  }
}

uint32_t orc_ammod_process(orc_ammod *h, const int16_t *pcm, uint32_t n, int8_t *iq_out)
{
  int16_t *m = (int16_t *)calloc((size_t)n + 1, sizeof(int16_t));
  for (uint32_t i = 0; i < n; i++)
  {
    float signal = (float)pcm[i] / 32768;                  /* :583 */
    signal *= h->index;
    signal = signal + 1;
    signal /= 2;
    m[i] = f2i16(signal * 128 * 250);                      /* :603-604: I and Q alike */
  }
  cascade_run(h->ip[0], m, n, iq_out, 0);
  cascade_run(h->ip[1], m, n, iq_out, 1);
  free(m);
  return n << 9;
}

/* FmModulator (FmModulator.cc): modulateSignal :586-627 -- per 8 kS/s sample the Nco (8000 Hz
 * sample rate, :221) gets frequency = deviation * pcm / 32768 and run() returns cos/sin of the
 * phase BEFORE the step (Nco.cc:186-199: sinf/cosf under the C++ overloads), scaled by 16000;
 * deviation 3500 Hz by default (:218); resetModulator (:277-300) leaves the Nco alone. */
struct orc_fmmod
{
  float deviation;
  float acc;
  q15i_t ip[2][8];
};

orc_fmmod *orc_fmmod_create(void)
{
  struct orc_fmmod *h = (struct orc_fmmod *)malloc(sizeof(*h));
  memset(h, 0, sizeof(*h));
  h->deviation = 3500;
  h->acc = 0;
  cascade_init(h->ip, AUDIO_D40);
  return h;
}

void orc_fmmod_destroy(orc_fmmod *h)
{
  free(h);
}

void orc_fmmod_reset(orc_fmmod *h)
{
  for (int r = 0; r < 2; r++)
  {
    for (int s = 0; s < 8; s++)
    {
      q15i_reset(&h->ip[r][s]);
    }
  }
}

void orc_fmmod_set_deviation(orc_fmmod *h, float deviation)
{
  /* FmModulator.cc:336-346 tests the CURRENT member, not the argument (kept as is) */
  if ((h->deviation >= 0) && (h->deviation <= 3500))
  {
    h->deviation = deviation;
  }
}

uint32_t orc_fmmod_process(orc_fmmod *h, const int16_t *pcm, uint32_t n, int8_t *iq_out)
{
  int16_t *mi = (int16_t *)calloc((size_t)n + 1, sizeof(int16_t));
  int16_t *mq = (int16_t *)calloc((size_t)n + 1, sizeof(int16_t));
  for (uint32_t i = 0; i < n; i++)
  {
    float f = h->deviation * (float)pcm[i] / 32768;        /* :597 */
    float step = (float)((2 * M_PI * f) / 8000.0f);        /* PhaseAccumulator.cc:105 */
    float phase = h->acc;                                  /* PhaseAccumulator.cc:157-181 */
    h->acc += step;
    while (h->acc > M_PI)
    {
      h->acc -= (2 * M_PI);
    }
    while (h->acc < (-M_PI))
    {
      h->acc += (2 * M_PI);
    }
    float iv = cosf(phase), qv = sinf(phase);
    iv *= 16000;
    qv *= 16000;
    mi[i] = f2i16(iv);
    mq[i] = f2i16(qv);
  }
  cascade_run(h->ip[0], mi, n, iq_out, 0);
  cascade_run(h->ip[1], mq, n, iq_out, 1);
  free(mi); free(mq);
  return n << 9;
}

/* WbFmModulator (WbFmModulator.cc): acceptData :341-356 = increasePcmSampleRate (:389-425: the
 * PCM through stages 1-5, x32) -> modulateSignal (:586-625: a 256 kS/s Nco, frequency =
 * deviation * x / 1024, runFast's table lookup, x900) -> increaseModulatedSampleRate (:447-492:
 * stages 6-8 on I and Q, x8, (int8_t)).  Deviation 70000 Hz by default (:182), setter range test
 * on the current member (:300-310, <= 112000); resetModulator leaves the Nco alone. */
struct orc_wbfmmod
{
  float deviation;
  float acc;
  q15i_t head[5];
  q15i_t tail[2][3];
  float sin_t[16384], cos_t[16384];
};

orc_wbfmmod *orc_wbfmmod_create(void)
{
  struct orc_wbfmmod *h = (struct orc_wbfmmod *)malloc(sizeof(*h));
  memset(h, 0, sizeof(*h));
  h->deviation = 70000;
  q15i_init(&h->head[0], AUDIO_D40, 40, 2);
  q15i_init(&h->head[1], INTERP_HB8, 8, 2);
  q15i_init(&h->head[2], INTERP_HB3, 4, 2);
  q15i_init(&h->head[3], INTERP_HB8, 8, 2);
  q15i_init(&h->head[4], INTERP_HB8, 8, 2);
  for (int r = 0; r < 2; r++)
  {
    q15i_init(&h->tail[r][0], INTERP_HB3, 4, 2);
    q15i_init(&h->tail[r][1], INTERP_HB2, 4, 2);
    q15i_init(&h->tail[r][2], INTERP_HB1, 4, 2);
  }
  /* Nco.cc:50-61 (same tables as orc_nco_create) */
  float inc = (float)(2 * M_PI / 16384);
  float ang = (float)(-M_PI);
  for (int i = 0; i < 16384; i++)
  {
    h->sin_t[i] = sinf(ang);
    h->cos_t[i] = cosf(ang);
    ang += inc;
  }
  return h;
}

void orc_wbfmmod_destroy(orc_wbfmmod *h)
{
  free(h);
}

void orc_wbfmmod_reset(orc_wbfmmod *h)
{
  for (int s = 0; s < 5; s++)
  {
    q15i_reset(&h->head[s]);
  }
  for (int r = 0; r < 2; r++)
  {
    for (int s = 0; s < 3; s++)
    {
      q15i_reset(&h->tail[r][s]);
    }
  }
}

void orc_wbfmmod_set_deviation(orc_wbfmmod *h, float deviation)
{
  if ((h->deviation >= 0) && (h->deviation <= 112000))
  {
    h->deviation = deviation;
  }
}

uint32_t orc_wbfmmod_process(orc_wbfmmod *h, const int16_t *pcm, uint32_t n, int8_t *iq_out)
{
  const size_t n32 = (size_t)n * 32;
  int16_t *a = (int16_t *)malloc((n32 * 8 + 2) * sizeof(int16_t));
  int16_t *b = (int16_t *)malloc((n32 * 8 + 2) * sizeof(int16_t));
  int16_t *mi = (int16_t *)malloc((n32 + 1) * sizeof(int16_t));
  int16_t *mq = (int16_t *)malloc((n32 + 1) * sizeof(int16_t));
  uint32_t cnt = n;
  memcpy(a, pcm, (size_t)n * sizeof(int16_t));
  for (int s = 0; s < 5; s++)
  {
    q15i_run(&h->head[s], a, cnt, b);
    cnt *= 2;
    int16_t *t = a; a = b; b = t;
  }
  for (size_t i = 0; i < n32; i++)
  {
    float f = h->deviation * (float)a[i] / 1024;           /* :601 */
    float step = (float)((2 * M_PI * f) / 256000.0f);      /* PhaseAccumulator.cc:105 */
    float phase = h->acc;
    h->acc += step;
    while (h->acc > M_PI)
    {
      h->acc -= (2 * M_PI);
    }
    while (h->acc < (-M_PI))
    {
      h->acc += (2 * M_PI);
    }
    int idx = d2i16(phase * 16384 / (2 * M_PI));          /* Nco.cc:231 */
    idx += 8192;
    if (idx < 0) idx = 0;
    else if (idx > 16383) idx = 16383;
    float iv = h->cos_t[idx], qv = h->sin_t[idx];
    iv *= 900;
    qv *= 900;
    mi[i] = f2i16(iv);
    mq[i] = f2i16(qv);
  }
  for (int r = 0; r < 2; r++)
  {
    cnt = (uint32_t)n32;
    memcpy(a, r ? mq : mi, n32 * sizeof(int16_t));
    for (int s = 0; s < 3; s++)
    {
      q15i_run(&h->tail[r][s], a, cnt, b);
      cnt *= 2;
      int16_t *t = a; a = b; b = t;
    }
    for (uint32_t i = 0; i < cnt; i++)
    {
      iq_out[2 * (size_t)i + r] = (int8_t)(uint8_t)((uint16_t)a[i] & 0xffu);
    }
  }
  free(a); free(b); free(mi); free(mq);
  return n << 9;                                           /* :354: (32 n) << 4 */
}

/* ------------------------------------------------------------------ tx PCM ring
 * BasebandDataProcessor's 16-slot ring of 512-sample PCM blocks between the PCM reader thread and
 * the transmit callback (BasebandDataProcessor.cc): getNextUnfilledBuffer :410-425 (writer),
 * getNextFilledBuffer :476-606 (reader with the pacing policy: lag > 10 drops a block, lag < 6
 * repeats one; first read after start() jumps to pcmReaderStartIndexTable[writer]), ctor :41-84,
 * start/stop :306-356 (Idle <-> Running; stop clears `synchronized`). */
#define ORC_RING 16
struct orc_txring
{
  uint32_t writer, reader;
  int running, synchronized;
  uint32_t produced, consumed, dropped, added;
  int16_t buf[ORC_RING][512];
};
static const int k_reader_start[ORC_RING] = {8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7};

orc_txring *orc_txring_create(void)
{
  struct orc_txring *h = (struct orc_txring *)calloc(1, sizeof(*h));
  h->writer = ORC_RING - 1;
  h->reader = (uint32_t)k_reader_start[h->writer];
  return h;
}

void orc_txring_destroy(orc_txring *h)
{
  free(h);
}

void orc_txring_set_running(orc_txring *h, int running)
{
  if (running)
  {
    h->running = 1;                                        /* start(): Idle -> Running (:306-315) */
  }
  else if (h->running)
  {
    h->running = 0;                                        /* stop(): Running -> Idle, unsynchronised (:341-354) */
    h->synchronized = 0;
  }
}

void orc_txring_write(orc_txring *h, const int16_t *pcm512)
{
  h->writer++;
  h->writer %= ORC_RING;
  memcpy(h->buf[h->writer], pcm512, 512 * sizeof(int16_t));
  h->produced++;
}

void orc_txring_read(orc_txring *h, int16_t *pcm512)
{
  int32_t u = (int32_t)h->writer, l = (int32_t)h->reader;
  if (u < l)
  {
    u += ORC_RING - 1;                                     /* :511-514 (sic: SIZE - 1) */
  }
  const int32_t lag = u - l;
  if (lag > 10)
  {
    h->reader++;
    h->reader %= ORC_RING;
    h->dropped++;
  }
  else if (lag < 6)
  {
    int32_t d = (int32_t)h->reader - 1;
    if (d < 0)
    {
      d += ORC_RING;
    }
    h->reader = (uint32_t)d;
    h->added++;
  }
  if (h->running)
  {
    if (!h->synchronized)
    {
      h->synchronized = 1;
      h->reader = (uint32_t)k_reader_start[h->writer];
    }
    memcpy(pcm512, h->buf[h->reader], 512 * sizeof(int16_t));
    h->reader++;
    h->reader %= ORC_RING;
    h->consumed++;
  }
  else
  {
    memset(pcm512, 0, 512 * sizeof(int16_t));              /* zeroPcmBuffer */
  }
}

void orc_txring_stats(const orc_txring *h, uint32_t *out6)
{
  out6[0] = h->produced; out6[1] = h->consumed; out6[2] = h->dropped; out6[3] = h->added;
  out6[4] = h->writer; out6[5] = h->reader;
}

/* ------------------------------------------------------------------ Nco */
struct orc_nco
{
  float sample_rate, frequency;
  float step, acc;
  float sin_t[16384], cos_t[16384];
};

static void nco_set_step(struct orc_nco *h)
{
  /* PhaseAccumulator.cc:41 / :105: double expression stored to float */
  h->step = (float)((2 * M_PI * h->frequency) / h->sample_rate);
}

orc_nco *orc_nco_create(float sample_rate, float frequency)
{
  struct orc_nco *h = (struct orc_nco *)malloc(sizeof(*h));
  /* Nco.cc:50-61: float phase accumulated by float increments; sin/cos of a
   * float argument resolve to sinf/cosf under the C++ overloads. */
  float inc = (float)(2 * M_PI / 16384);
  float ang = (float)(-M_PI);
  for (int i = 0; i < 16384; i++)
  {
    h->sin_t[i] = sinf(ang);
    h->cos_t[i] = cosf(ang);
    ang += inc;
  }
  h->sample_rate = sample_rate;
  h->frequency = frequency;
  nco_set_step(h);
  h->acc = 0;
  return h;
}

void orc_nco_destroy(orc_nco *h)
{
  free(h);
}

void orc_nco_set_frequency(orc_nco *h, float frequency)
{
  h->frequency = frequency;
  nco_set_step(h);
}

void orc_nco_reset(orc_nco *h)
{
  h->acc = 0;
}

/* PhaseAccumulator::run (:157-181) */
static inline float phase_run(struct orc_nco *h)
{
  float phase = h->acc;
  h->acc += h->step;
  while (h->acc > M_PI)
  {
    h->acc = (float)((double)h->acc - (2 * M_PI));
  }
  while (h->acc < (-M_PI))
  {
    h->acc = (float)((double)h->acc + (2 * M_PI));
  }
  return phase;
}

void orc_nco_run(orc_nco *h, int fast, uint32_t count, float *i_out, float *q_out)
{
  for (uint32_t k = 0; k < count; k++)
  {
    float phase = phase_run(h);
    if (fast)
    {
      /* Nco::runFast (:222-257) */
      int idx = (int)d2i16((double)(phase * 16384) / (2 * M_PI));
      idx += 8192;
      if (idx < 0)
      {
        idx = 0;
      }
      else if (idx > 16383)
      {
        idx = 16383;
      }
      i_out[k] = h->cos_t[idx];
      q_out[k] = h->sin_t[idx];
    }
    else
    {
      /* Nco::run (:186-199) */
      i_out[k] = cosf(phase);
      q_out[k] = sinf(phase);
    }
  }
}

void orc_nco_tables(orc_nco *h, float *sin_out, float *cos_out)
{
  memcpy(sin_out, h->sin_t, sizeof(h->sin_t));
  memcpy(cos_out, h->cos_t, sizeof(h->cos_t));
}

/* ------------------------------------------------------------------ primitives */
void orc_quantise(const float *coeffs, int count, int16_t *out)
{
  quantise(coeffs, count, out);
}

uint32_t orc_decimate(const float *coeffs, int taps, int factor,
                      const int16_t *in, uint32_t count, int16_t *out)
{
  q15_t s;
  q15_init(&s, coeffs, taps, factor);
  return q15_run(&s, in, count, out);
}

void orc_interpolate(const float *coeffs, int taps, int factor,
                     const int16_t *in, uint32_t count, int16_t *out)
{
  q15i_t s;
  q15i_init(&s, coeffs, taps, factor);
  q15i_run(&s, in, count, out);
}

/* FirFilter over a buffer (FirFilter.cc:144-185), from a cleared state. */
void orc_fir(const float *h, int n, const float *in, uint32_t count, float *out)
{
  firf_t s;
  firf_init(&s, h, n);
  for (uint32_t k = 0; k < count; k++)
  {
    out[k] = firf_step(&s, in[k]);
  }
}

void orc_iir(const float *b, int nb, const float *a, int na,
             const float *in, uint32_t count, float *out)
{
  iirf_t s;
  iirf_init(&s, b, nb, a, na);
  for (uint32_t k = 0; k < count; k++)
  {
    out[k] = iirf_step(&s, in[k]);
  }
}

int16_t orc_float_to_int16(float v)
{
  return f2i16(v);
}

#define TBL(nm) { #nm, nm, (int)(sizeof(nm) / sizeof(nm[0])) }
static const struct { const char *name; const float *data; int n; } g_named[] =
{
  TBL(HB1), TBL(HB2), TBL(HB3), TBL(WBFM_D1), TBL(POST_D12), TBL(AUDIO_D40),
  TBL(DEEMPH_B), TBL(DEEMPH_A), TBL(FM_TUNER_D32), TBL(FM_DIFF), TBL(AM_D1),
  TBL(AM_D2), TBL(AM_D3), TBL(DCREM_B), TBL(DCREM_A), TBL(SSB_DELAY),
  TBL(SSB_HILBERT), TBL(INTERP_HB8), TBL(INTERP_HB3), TBL(INTERP_HB2),
  TBL(INTERP_HB1), TBL(INTERPSIG_S1)
};

int orc_table(const char *name, float *out, int cap)
{
  for (size_t i = 0; i < sizeof(g_named) / sizeof(g_named[0]); i++)
  {
    if (strcmp(name, g_named[i].name) == 0)
    {
      int n = g_named[i].n < cap ? g_named[i].n : cap;
      memcpy(out, g_named[i].data, (size_t)n * sizeof(float));
      return g_named[i].n;
    }
  }
  return 0;
}

/* ---- signals/{am,dsb,pm,fm}.cc ------------------------------------------------------------
 * Each tool reads int16 PCM from stdin and writes int16 (I,Q) pairs to stdout, float arithmetic.
 * The C++ sources include <math.h> under `using namespace std`, so cos/sin of a float argument
 * are the float overloads (cosf/sinf), as SURVEY 8(a) N2 found for Nco::run; "x *= 0.8" and
 * "x *= M_PI" multiply in double and round back to float. */
void orc_siggen(int kind, const int16_t *pcm, uint32_t n, int16_t *iq_pairs, float *theta_io)
{
  float theta = theta_io ? *theta_io : 0.0f;
  for (uint32_t k = 0; k < n; k++)
  {
    float s = (float)pcm[k];
    int16_t vi, vq;
    if (kind == 0)
    {                                   /* am.cc:44-49 */
      s = (float)((double)s * 0.8);
      s = s + 65536;
      s = s / 4;
      vi = vq = (int16_t)s;
    }
    else if (kind == 1)
    {                                   /* dsb.cc:42-45 */
      s = s / 4;
      vi = vq = (int16_t)s;
    }
    else if (kind == 2)
    {                                   /* pm.cc:45-52 */
      s = s / 60000;
      s = (float)((double)s * M_PI);
      const float i = cosf(s) * 16000;
      const float q = sinf(s) * 16000;
      vi = (int16_t)i;
      vq = (int16_t)q;
    }
    else
    {                                   /* fm.cc:53-74, kF = 3.5 */
      float tn = (float)pcm[k];
      tn = tn / 65536;
      tn *= 3.5f;
      theta = theta + tn;
      while (theta > (2 * M_PI))
      {
        theta = (float)(theta - (2 * M_PI));
      }
      while (theta < (-(2 * M_PI)))
      {
        theta = (float)(theta + (2 * M_PI));
      }
      const float i = cosf(theta) * 16000;
      const float q = sinf(theta) * 16000;
      vi = (int16_t)i;
      vq = (int16_t)q;
    }
    iq_pairs[2 * k] = vi;
    iq_pairs[2 * k + 1] = vq;
  }
  if (theta_io)
  {
    *theta_io = theta;
  }
}
