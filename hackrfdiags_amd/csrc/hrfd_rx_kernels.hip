// hackrfdiags_amd/csrc/hrfd_rx_kernels.hip -- gfx950 (MI355X, CDNA4) receive kernels.
//
// One fused kernel per demodulator mode.  A workgroup (8 wave64) owns one block
// of one channel: 262144 bytes of int8 IQ at 2.048 MS/s in, 512 int16 PCM out,
// nothing in between ever leaves the CU.
//
//   phase A  (all waves)  coalesced 16-byte loads, one lane = 8 IQ samples in =
//            one 256 kS/s sample out: three half-band /2 stages on packed (I,Q)
//            int16 pairs, Fs/4 rotation, squelch magnitude, atan2 table lookup,
//            phase difference, +-pi wrap, de-emphasis numerator  -> v[n] in LDS
//            mirrors IqDataProcessor::reduceSampleRate (IqDataProcessor.cc:429-500),
//            upconvertByFsOver4 (:771-815), SignalDetector::detectSignal
//            (SignalDetector.cc:205-274), WbFmDemodulator::demodulateSignal
//            (WbFmDemodulator.cc:381-439) up to the recursive part.
//   phase B  (one wave)   the float recurrence y[n] = v[n] - a1*y[n-1]
//            (IirFilter.cc:161-176) on 64 time tiles at once, each warmed up on
//            384 samples of history and verified bit-for-bit against its left
//            neighbour; a miss is counted, the call's state is not committed and
//            the host replays it on the exact one-lane path.
//   phase C  (all waves)  (int16) narrowing with x86 semantics, D(8,4), D(12,4),
//            D(40,2) with v_dot2_i32_i16 from LDS -> PCM
//            (WbFmDemodulator::createPcmData, WbFmDemodulator.cc:460-500;
//             Decimator_int16::filterData, Decimator_int16.cc:176-249).
//
// Arithmetic contract: every integer stage is bit-exact (int32 wrap-around
// accumulate, arithmetic >>15, low-16 narrowing); every float operation is a
// single IEEE round-to-nearest op in the reference's order (this file is built
// with -ffp-contract=off; no fma, no fast-math).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hrfd_device.h"
#include "hrfd_tables.h"

namespace hrfd {

typedef short s2 __attribute__((ext_vector_type(2)));    // one (I,Q) pair or two taps

__device__ __forceinline__ s2 as_s2(uint32_t u) { return __builtin_bit_cast(s2, u); }
__device__ __forceinline__ uint32_t as_u32(s2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// lane i <- lane i-1 of `v`; lane 0 <- lane 0 of `carry` (v_mov_b32_dpp wave_shr:1)
__device__ __forceinline__ uint32_t shr1(uint32_t v, uint32_t carry)
{
  return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t lane63(uint32_t v)
{
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// ---- half-band /2 stage on packed (I,Q), offset-binary domain ---------------
// Reference: y = (16384 + h0*(a+c) + 16384*b) >> 15 with h0 = 8192 + d
// (Decimator_int16.cc:176-249 with the 3-tap tables of IqDataProcessor.cc:8-27).
// With a' = a+128 etc. and T = a'+c':  y' = y+128 = (T + 2b' + ((D*T + K) >> SH)) >> 2,
// where (D*T + K) >> SH == floor((d*(T-256) + 16384) / 8192).  Every intermediate
// fits int16, so the stage runs on v_pk_* with I and Q in the two halves.
// Checked exhaustively against the direct form in tests/test_frontend_formula.py.
template <int D, int K, int SH>
__device__ __forceinline__ s2 halfband(s2 a, s2 b, s2 c)
{
  s2 t = a + c;
  s2 k = (t * (short)D + (short)K) >> (short)SH;
  return (t + b + b + k) >> (short)2;
}
#define HB1(a, b, c) halfband<14, 12800, 13>(a, b, c)   /* h0 = 8206 */
#define HB2(a, b, c) halfband<57, 1792, 13>(a, b, c)    /* h0 = 8249 */
#define HB3(a, b, c) halfband<29, -5376, 10>(a, b, c)   /* h0 = 8424 = 8192 + 8*29 */

// Carry between consecutive 1 KiB chunks of one wave (values of the last lane).
struct FeCarry
{
  uint32_t x7;    // last input pair
  uint32_t y13;   // last stage-1 output pair
  uint32_t y21;   // last stage-2 output pair
};

// 16 raw bytes = 8 IQ samples -> one 256 kS/s sample (offset-binary, low byte
// of each half = value + 128).  `cin` supplies lane 0's left neighbour.
__device__ __forceinline__ uint32_t frontend(const uint4 raw, const FeCarry cin, FeCarry &cout)
{
  const uint32_t r0 = raw.x ^ 0x80808080u, r1 = raw.y ^ 0x80808080u;
  const uint32_t r2 = raw.z ^ 0x80808080u, r3 = raw.w ^ 0x80808080u;
  // zero-extend bytes (I0,Q0,I1,Q1) into two packed pairs
  const s2 x0 = as_s2(__builtin_amdgcn_perm(0u, r0, 0x0c010c00u));
  const s2 x1 = as_s2(__builtin_amdgcn_perm(0u, r0, 0x0c030c02u));
  const s2 x2 = as_s2(__builtin_amdgcn_perm(0u, r1, 0x0c010c00u));
  const s2 x3 = as_s2(__builtin_amdgcn_perm(0u, r1, 0x0c030c02u));
  const s2 x4 = as_s2(__builtin_amdgcn_perm(0u, r2, 0x0c010c00u));
  const s2 x5 = as_s2(__builtin_amdgcn_perm(0u, r2, 0x0c030c02u));
  const s2 x6 = as_s2(__builtin_amdgcn_perm(0u, r3, 0x0c010c00u));
  const s2 x7 = as_s2(__builtin_amdgcn_perm(0u, r3, 0x0c030c02u));

  const s2 xm1 = as_s2(shr1(as_u32(x7), cin.x7));
  const s2 y10 = HB1(xm1, x0, x1);
  const s2 y11 = HB1(x1, x2, x3);
  const s2 y12 = HB1(x3, x4, x5);
  const s2 y13 = HB1(x5, x6, x7);

  const s2 y1m1 = as_s2(shr1(as_u32(y13), cin.y13));
  const s2 y20 = HB2(y1m1, y10, y11);
  const s2 y21 = HB2(y11, y12, y13);

  const s2 y2m1 = as_s2(shr1(as_u32(y21), cin.y21));
  const s2 y3 = HB3(y2m1, y20, y21);

  cout.x7 = lane63(as_u32(x7));
  cout.y13 = lane63(as_u32(y13));
  cout.y21 = lane63(as_u32(y21));
  return as_u32(y3);
}

// The carry that a chunk boundary needs is a function of the 16 bytes before
// it only (x1..x7 of that slot): recompute it from those bytes.
__device__ __forceinline__ FeCarry carry_from_16(const uint4 raw)
{
  const uint32_t r0 = raw.x ^ 0x80808080u, r1 = raw.y ^ 0x80808080u;
  const uint32_t r2 = raw.z ^ 0x80808080u, r3 = raw.w ^ 0x80808080u;
  const s2 x1 = as_s2(__builtin_amdgcn_perm(0u, r0, 0x0c030c02u));
  const s2 x2 = as_s2(__builtin_amdgcn_perm(0u, r1, 0x0c010c00u));
  const s2 x3 = as_s2(__builtin_amdgcn_perm(0u, r1, 0x0c030c02u));
  const s2 x4 = as_s2(__builtin_amdgcn_perm(0u, r2, 0x0c010c00u));
  const s2 x5 = as_s2(__builtin_amdgcn_perm(0u, r2, 0x0c030c02u));
  const s2 x6 = as_s2(__builtin_amdgcn_perm(0u, r3, 0x0c010c00u));
  const s2 x7 = as_s2(__builtin_amdgcn_perm(0u, r3, 0x0c030c02u));
  const s2 y11 = HB1(x1, x2, x3);
  const s2 y12 = HB1(x3, x4, x5);
  const s2 y13 = HB1(x5, x6, x7);
  FeCarry c;
  c.x7 = as_u32(x7);
  c.y13 = as_u32(y13);
  c.y21 = as_u32(HB2(y11, y12, y13));
  return c;
}

// ---- Fs/4 rotation + table index + squelch magnitude -------------------------
// y3 holds (I+128, Q+128) (low bytes significant: the (int8_t) narrowing of
// IqDataProcessor.cc:458,489).  upconvertByFsOver4 (:771-815) multiplies by
// {1, j, -1, -j}: rot 0 (I,Q), 1 (-Q,I), 2 (-I,-Q), 3 (Q,-I), int8 negation
// wrapping.  In index form (value+128) negation is (256 - idx) & 255.
struct Mixed
{
  uint32_t i_idx, q_idx;   // (uint8)(I+128), (uint8)(Q+128) after the mix: LUT indices
  uint32_t mag;            // max(|I|,|Q|) + (min(|I|,|Q|) >> 1), SignalDetector.cc:226-241
};

__device__ __forceinline__ Mixed mix_fs4(uint32_t y3, int rot)
{
  const uint32_t ui = y3 & 0xffu, uq = (y3 >> 16) & 0xffu;
  const bool swap = (rot & 1) != 0;
  const uint32_t a = swap ? uq : ui;
  const uint32_t b = swap ? ui : uq;
  const bool nega = (rot == 1) || (rot == 2);
  const bool negb = (rot == 2) || (rot == 3);
  Mixed m;
  m.i_idx = nega ? ((0u - a) & 0xffu) : a;
  m.q_idx = negb ? ((0u - b) & 0xffu) : b;
  // |v| of the int8 value v = idx - 128 (|-128| = 128 fits the reference's uint8)
  const int ai = abs((int)ui - 128), aq = abs((int)uq - 128);
  const int mx = max(ai, aq), mn = min(ai, aq);
  m.mag = (uint32_t)(mx + (mn >> 1));
  return m;
}

// deltaTheta wrap (WbFmDemodulator.cc:417-425).  The reference compares the
// float against the double M_PI: (double)d > M_PI  <=>  d >= 0x1.921fb6p+1f,
// and subtracts 2*M_PI in double before rounding back to float.
__device__ __forceinline__ float wrap_pi(float d)
{
  const float pi_up = 3.14159274101257324e+00f;       // smallest float > M_PI
  const double two_pi = 6.283185307179586476925286766559;
  if (d >= pi_up)
  {
    d = (float)((double)d - two_pi);
  }
  if (d <= -pi_up)
  {
    d = (float)((double)d + two_pi);
  }
  return d;
}

// (int16_t)f the way x86-64 does it (cvttss2si, then the low 16 bits): NaN and
// |f| >= 2^31 give 0x80000000 -> 0.  v_cvt_i32_f32 saturates instead, so the
// positive overflow is patched (INT_MAX can only come from saturation).
__device__ __forceinline__ int f2i16(float f)
{
  int v = (int)f;                       // v_cvt_i32_f32: trunc, saturating, NaN -> 0
  v = (v == 0x7fffffff) ? 0 : v;        // low 16 bits of 0x80000000
  return (int)(short)v;
}

// Q15 round/shift/narrow: (int16)((acc) >> 15), acc already includes 1<<14.
__device__ __forceinline__ int q15_out(int acc) { return (int)(short)(acc >> 15); }

__device__ __forceinline__ int dot2(uint32_t a, uint32_t taps, int acc)
{
  return __builtin_amdgcn_sdot2(as_s2(a), as_s2(taps), acc, false);
}

// packed, time-reversed taps: pair j holds (h[N-1-2j], h[N-2-2j]) so that
// sum_j dot2(x[base+2j .. base+2j+1], rtaps[j]) == sum_k h[k]*x[base+N-1-k].
template <int N>
struct RevTaps
{
  uint32_t p[N / 2];
  constexpr RevTaps(const int16_t (&h)[N]) : p{}
  {
    for (int j = 0; j < N / 2; j++)
    {
      const uint32_t lo = (uint16_t)h[N - 1 - 2 * j];
      const uint32_t hi = (uint16_t)h[N - 2 - 2 * j];
      p[j] = lo | (hi << 16);
    }
  }
};
__constant__ constexpr RevTaps<N_WBFM_D1> kRevWbD1(Q_WBFM_D1);
__constant__ constexpr RevTaps<N_POST_D12> kRevD12(Q_POST_D12);
__constant__ constexpr RevTaps<N_AUDIO_D40> kRevD40(Q_AUDIO_D40);

// ---- workgroup -> (channel, block) mapping -----------------------------------
// Workgroups are dealt round-robin over the 8 XCDs, so ids w and w+8 share an
// L2.  Give each XCD whole channels and walk a channel's blocks consecutively:
// the history a block re-reads from its predecessor's tail is then warm in the
// same L2.  Placement affects speed only (MI355X_MICROARCH, "Workgroup dispatch").
__device__ __forceinline__ bool map_unit(uint32_t w, uint32_t n_list, uint32_t n_blocks,
                                         uint32_t &ci, uint32_t &b)
{
  const uint32_t xcd = w & 7u, q = w >> 3;
  const uint32_t cc = q / n_blocks;
  b = q - cc * n_blocks;
  ci = cc * 8u + xcd;
  return ci < n_list;
}

// =============================================================================
//  WBFM  (mode 3)  and  NONE (mode 0: front end + squelch only)
// =============================================================================
// LDS map (dword indices into lds[]):
//   phase A/B   v/y stream       index = pos + hal            pos in [-hal, n256)
//   phase C     S  int16 pairs   dword (pos + kHist) / 2      pos in [-kHist, n256)
//               U  int16         kUOff*2 + (m + 160)          m   in [-160, n256/4)
//               V  int16         kVOff*2 + (k + 38)           k   in [-38, n256/16)
constexpr int kSDwords = (kMaxN256 + kHist) / 2;             // 8544
constexpr int kUOff = 8576;                                   // dword offset of U (>= kSDwords, 16-B aligned)
constexpr int kUHist = 160;
constexpr int kVOff = kUOff + (kMaxN256 / 4 + kUHist) / 2;    // 10704
constexpr int kVHist = 38;
constexpr int kPairsPerThread = (kMaxN256 + kHist) / 2 / kThreads + 1;   // 17
static_assert(kVOff + (kMaxN256 / 16 + kVHist) / 2 + 1 <= kMaxNV, "LDS map");
static_assert(kSDwords <= kUOff && (kUOff % 4) == 0, "LDS map");

// Everything phase A needs to (re)produce a range of the 256 kS/s stream.
struct StreamCtx
{
  const RxParams *P;
  const int8_t *blk;        // first raw byte of this block
  const ChanState *st;
  uint32_t *lds;
  size_t ounit;
  float kgain;
  int hal, vstart, n256;
  int lane;
  bool first;
};

// Run chunks [c0, c1) (64 samples each, chunk c covers positions vstart + 64c ..)
// on one wave and store v[pos] for pos in [wlo, whi).  `side` enables the
// once-only side outputs (squelch magnitude, optional 256 kS/s dump).
// Returns theta and b0*x of the last sample through c_theta / c_p.
template <int MODE>
__device__ __forceinline__ void produce_stream(const StreamCtx &X, const int c0, const int c1,
                                               const int wlo, const int whi, const bool side,
                                               uint32_t &magsum, uint32_t &c_theta, uint32_t &c_p)
{
  const RxParams &P = *X.P;
  const int lane = X.lane;
  const int rot = lane & 3;                              // position & 3 (chunks are 64-aligned)
  FeCarry fc = {0x00800080u, 0x00800080u, 0x00800080u};
  bool have_carry = false;
  c_theta = 0;
  c_p = 0;
  if (X.first && c0 == 0)
  {
    // the stream continues from the previous call: carried state
    fc = carry_from_16(*reinterpret_cast<const uint4 *>(X.st->fe_tail));
    c_theta = f2u(X.st->wb_theta);
    c_p = f2u(X.st->wb_p);
    have_carry = true;
  }
  // Without carried state one extra, discarded, chunk in front re-creates the
  // carries (their garbage-in only reaches the discarded chunk's lanes 0..2).
  const int cbeg = have_carry ? c0 : c0 - 1;
  const uint4 *src =
      reinterpret_cast<const uint4 *>(X.blk + ((int64_t)X.vstart + 64 * (int64_t)cbeg) * 16) + lane;
  // software pipeline: raw loads run two chunks ahead of their use
  uint4 raw0 = src[0];
  uint4 raw1 = (cbeg + 1 < c1) ? src[64] : raw0;
  for (int ch = cbeg; ch < c1; ch++)
  {
    const uint4 raw = raw0;
    raw0 = raw1;
    if (ch + 2 < c1)
    {
      raw1 = src[(size_t)(ch + 2 - cbeg) * 64];
    }
    FeCarry fo;
    const uint32_t y3 = frontend(raw, fc, fo);
    fc = fo;
    const Mixed m = mix_fs4(y3, rot);
    const int pos = X.vstart + 64 * ch + lane;
    const bool live = (ch >= c0);                        // false only for the discarded chunk
    if (side && live && pos >= 0)
    {
      magsum += m.mag;
      if (P.iq256 != nullptr)
      {
        const uint16_t pair = (uint16_t)((m.i_idx ^ 0x80u) | ((m.q_idx ^ 0x80u) << 8));
        reinterpret_cast<uint16_t *>(P.iq256 + X.ounit * (size_t)(2 * X.n256))[pos] = pair;
      }
    }
    if (MODE == 3)
    {
      const float theta = P.atan2_lut[(m.q_idx << 8) | m.i_idx];
      const float thp = u2f(shr1(f2u(theta), c_theta));
      float d = theta - thp;
      d = wrap_pi(d);
      const float x = X.kgain * d;
      const float p = DEEMPH_B0 * x;                     // b0*x[n]; b1 == b0: also the next b1*x[n-1]
      const float pp = u2f(shr1(f2u(p), c_p));
      const float v = p + pp;
      c_theta = lane63(f2u(theta));
      c_p = lane63(f2u(p));
      if (live && pos >= wlo && pos < whi)
      {
        X.lds[pos + X.hal] = f2u(v);
      }
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void k_rx_wbfm(const RxParams P)
{
  __shared__ __attribute__((aligned(16))) uint32_t lds[kMaxNV];
  __shared__ uint32_t red[kWaves];
  __shared__ float tailcarry[2];        // theta, b0*x of the block's last sample

  uint32_t ci, b;
  if (!map_unit(blockIdx.x, P.n_list, P.n_blocks, ci, b))
  {
    return;
  }
  const uint32_t c = P.chan_list[ci];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int n256 = (int)P.n256;
  const bool first = (b == 0);
  const int hal = P.hal;
  const ChanState *st = P.state + c;
  const ChanCfg cfg = P.cfg[c];
  const size_t unit = (size_t)c * P.n_blocks + b;                    // launch-local scratch index

  StreamCtx X;
  X.P = &P;
  X.blk = P.iq + (uint64_t)c * P.ch_stride + (uint64_t)b * P.block_bytes;
  X.st = st;
  X.lds = lds;
  X.ounit = (size_t)c * P.out_blocks + P.out_b0 + b;                 // index in the caller's outputs
  // K = (gain/75000)*32767 in float, that order (WbFmDemodulator.cc:392-395)
  X.kgain = cfg.gain_wbfm / 75000.0f;
  X.kgain = X.kgain * 32767.0f;
  X.hal = hal;
  X.vstart = first ? 0 : -hal;                           // first position of v we produce
  X.n256 = n256;
  X.lane = lane;
  X.first = first;

  // ----------------------------------------------------------------- phase A
  const int nch = (n256 - X.vstart) >> 6;                // 1 KiB chunks to run
  const int cpw = (nch + kWaves - 1) / kWaves;
  const int c0 = wave * cpw;
  const int c1 = min(nch, c0 + cpw);
  uint32_t magsum = 0;
  if (c0 < c1)
  {
    uint32_t c_theta, c_p;
    produce_stream<MODE>(X, c0, c1, X.vstart, n256, true, magsum, c_theta, c_p);
    if (MODE == 3 && c1 == nch && lane == 0)
    {
      tailcarry[0] = u2f(c_theta);
      tailcarry[1] = u2f(c_p);
    }
  }

  // block-mean magnitude: wave reduce, then across waves
  for (int off = 32; off > 0; off >>= 1)
  {
    magsum += __shfl_down(magsum, off);
  }
  if (lane == 0)
  {
    red[wave] = magsum;
  }
  __syncthreads();
  uint32_t total = 0;
  for (int w = 0; w < kWaves; w++)
  {
    total += red[w];
  }
  const uint32_t mean_mag = total / (uint32_t)n256;      // SignalDetector.cc:255
  // DbfsCalculator::convertMagnitudeToDbFs (:111-147) with a 7-bit full scale
  int32_t dbfs = P.dbfs[min(mean_mag, 127u)] - 42;
  dbfs = (int32_t)((uint32_t)dbfs - P.gain_db);
  const bool present = dbfs >= cfg.threshold;
  // Squelch::run + SignalTracker::run: allowed = present || tracking.  For b > 0
  // the predecessor's `present` is not known here: the batch speculates "open"
  // and k_rx_epilogue verifies it.
  const bool allowed = first ? (present || st->tracking != 0) : true;
  if (tid == 0)
  {
    P.magnitude[X.ounit] = mean_mag;
    P.present[unit] = present ? 1 : 0;
  }

  const bool last = (b + 1 == P.n_blocks);
  ChanState *so = P.state_out + c;
  if (last && tid < 4)
  {
    // front-end carry for the next call: the last 16 raw bytes of this block
    reinterpret_cast<uint32_t *>(so->fe_tail)[tid] =
        reinterpret_cast<const uint32_t *>(X.blk + P.block_bytes - 16)[tid];
  }
  if (MODE != 3 || !allowed)
  {
    return;                                              // demodulator untouched (state frozen)
  }

  // ----------------------------------------------------------------- phase B
  // y[n] = v[n] - a1*y[n-1] (IirFilter.cc:161-176 with one recursive tap), in place.
  // 64 tiles of T samples end at n256; tile i = [origin + i*T, origin + (i+1)*T).
  // Each lane first runs `warm` samples of history from y = 0 (from the carried
  // y when the stream start lies inside its window), then its own tile.  A lane
  // whose warmed-up y[s-1] is not bit-identical to its left neighbour's final y
  // has not re-synchronised: its tile is re-derived and re-run from the true
  // value (rare: DESIGN.md gives the measured rates).  Tile 0 is sacrificial (it
  // lies before the first history sample anybody reads) so that the chain is
  // anchored on a lane with warm + T samples behind it.
  const int T = P.tile;
  const int origin = P.origin;
  const int warm = P.warm;
  if (wave == 0)
  {
    const float a1 = DEEMPH_A1;
    if (P.serial)
    {
      // exact replay path (n_blocks == 1): one lane, the whole block in order
      if (lane == 0)
      {
        float y = st->wb_y;
        for (int n = 0; n < n256; n++)
        {
          const float r = a1 * y;
          y = u2f(lds[n + hal]) - r;
          lds[n + hal] = f2u(y);
        }
      }
    }
    else
    {
      const int s = origin + lane * T;
      float y = first ? st->wb_y : 0.0f;
      const uint32_t *vp = lds + (s - warm + hal);       // lane stride T is odd: bank-conflict free
      for (int k = 0; k < warm; k++)
      {
        const int n = s - warm + k;
        if (!first || n >= 0)
        {
          const float r = a1 * y;
          y = u2f(vp[k]) - r;
        }
      }
      const float y_spec = y;                            // speculated y[s-1]
      uint32_t *yp = lds + (s + hal);
      for (int k = 0; k < T; k++)
      {
        const int n = s + k;
        if (!first || n >= 0)
        {
          const float r = a1 * y;
          y = u2f(yp[k]) - r;
          yp[k] = f2u(y);
        }
      }
      // Anchors: lane 0 (tile 0 is sacrificial; the chain behind it is checked
      // across blocks by k_rx_epilogue) and, in a first block, lanes whose
      // window contains the true stream state (s - warm <= 0).
      const bool anchored = (lane == 0) || (first && (s - warm) <= 0);
      const float y_left = u2f(shr1(f2u(y), f2u(y_spec)));
      unsigned long long bad = __ballot(!anchored && (f2u(y_left) != f2u(y_spec)));
      uint32_t repairs = 0;
      while (bad != 0ull)
      {
        const int j = __ffsll((long long)bad) - 1;       // wave-uniform, >= 1
        bad &= ~(1ull << j);
        repairs++;
        const int sj = origin + j * T;
        // re-derive v over tile j (it was overwritten by the mis-started y)
        const int rc0 = (sj - X.vstart) >> 6;
        const int rc1 = (sj + T - X.vstart + 63) >> 6;
        uint32_t dummy_mag = 0, dummy_t, dummy_p;
        produce_stream<MODE>(X, rc0, rc1, sj, sj + T, false, dummy_mag, dummy_t, dummy_p);
        // re-run the tile from the true y[sj - 1] = final y of lane j-1
        const float y_true = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(y), j - 1));
        if (lane == j)
        {
          float yy = y_true;
          uint32_t *rp = lds + (sj + hal);
          for (int k = 0; k < T; k++)
          {
            const float r = a1 * yy;
            yy = u2f(rp[k]) - r;
            rp[k] = f2u(yy);
          }
          y = yy;
        }
        // the right neighbour's speculation must now match the corrected final y
        if (j + 1 < 64)
        {
          const uint32_t yj = (uint32_t)__builtin_amdgcn_readlane((int)f2u(y), j);
          const uint32_t sp = (uint32_t)__builtin_amdgcn_readlane((int)f2u(y_spec), j + 1);
          const int s1 = origin + (j + 1) * T;
          const bool anch1 = first && (s1 - warm) <= 0;
          if (!anch1 && yj != sp)
          {
            bad |= 1ull << (j + 1);
          }
        }
      }
      if (repairs != 0 && lane == 0)
      {
        atomicAdd(&P.counters[kCntRepair], repairs);
      }
    }
  }
  __syncthreads();
  if (tid == 0)
  {
    // cross-block check values: position -645 precedes every history sample the
    // integer stages read (-644) and lies in tile 1.
    const int chk = -kHist + 59;
    P.chk_spec[unit] = first ? 0.0f : u2f(lds[chk + hal]);
    P.chk_pub[unit] = u2f(lds[n256 + chk + hal]);
  }

  // ----------------------------------------------------------------- phase C
  // C1: s[n] = (int16_t)y[n] (WbFmDemodulator.cc:476), repacked in place as
  // int16 pairs at the bottom of the buffer: all reads, barrier, all writes.
  const int smin = first ? 0 : -kHist;
  const int npairs = (n256 - smin) >> 1;
  uint32_t packed[kPairsPerThread];
#pragma unroll
  for (int r = 0; r < kPairsPerThread; r++)
  {
    const int q = tid + r * kThreads;
    uint32_t w = 0;
    if (q < npairs)
    {
      const int pos = smin + 2 * q;
      const float ya = u2f(lds[pos + hal]);
      const float yb = u2f(lds[pos + 1 + hal]);
      w = ((uint32_t)f2i16(ya) & 0xffffu) | ((uint32_t)f2i16(yb) << 16);
      if (last && pos + 2 == n256)
      {
        so->wb_y = yb;
        so->wb_theta = tailcarry[0];
        so->wb_p = tailcarry[1];
      }
    }
    packed[r] = w;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kPairsPerThread; r++)
  {
    const int q = tid + r * kThreads;
    if (q < npairs)
    {
      lds[((smin + kHist) >> 1) + q] = packed[r];
    }
  }
  uint16_t *U16 = reinterpret_cast<uint16_t *>(lds + kUOff);
  uint16_t *V16 = reinterpret_cast<uint16_t *>(lds + kVOff);
  if (first)
  {
    // histories of the three integer stages from the carried state
    if (tid < kWbS / 2)
    {
      lds[((kHist - kWbS) >> 1) + tid] = reinterpret_cast<const uint32_t *>(st->wb_s)[tid];
    }
    if (tid < kWbU)
    {
      U16[kUHist - kWbU + tid] = (uint16_t)st->wb_u[tid];
    }
    if (tid < kWbV)
    {
      V16[kVHist - kWbV + tid] = (uint16_t)st->wb_v[tid];
    }
  }
  __syncthreads();

  // C2: U[m] = D(8,4)(S), WbFmDemodulator.cc:468-472; two outputs per thread
  {
    const int mmin = first ? 0 : -kUHist;
    const int nU = (n256 >> 2) - mmin;
    for (int q = tid; q < (nU >> 1); q += kThreads)
    {
      const int m = mmin + 2 * q;
      // S[4m-4 .. 4m+7] -> dwords (4m - 4 + kHist)/2 ...
      const uint32_t *sp = lds + ((4 * m - 4 + kHist) >> 1);
      const uint2 a = *reinterpret_cast<const uint2 *>(sp);
      const uint2 bq = *reinterpret_cast<const uint2 *>(sp + 2);
      const uint2 cq = *reinterpret_cast<const uint2 *>(sp + 4);
      int acc0 = 1 << 14, acc1 = 1 << 14;
      acc0 = dot2(a.x, kRevWbD1.p[0], acc0);
      acc0 = dot2(a.y, kRevWbD1.p[1], acc0);
      acc0 = dot2(bq.x, kRevWbD1.p[2], acc0);
      acc0 = dot2(bq.y, kRevWbD1.p[3], acc0);
      acc1 = dot2(bq.x, kRevWbD1.p[0], acc1);
      acc1 = dot2(bq.y, kRevWbD1.p[1], acc1);
      acc1 = dot2(cq.x, kRevWbD1.p[2], acc1);
      acc1 = dot2(cq.y, kRevWbD1.p[3], acc1);
      const uint32_t w = ((uint32_t)q15_out(acc0) & 0xffffu) | ((uint32_t)q15_out(acc1) << 16);
      lds[kUOff + ((m + kUHist) >> 1)] = w;
    }
  }
  __syncthreads();

  // C3: V[k] = D(12,4)(U); two outputs per thread
  {
    const int kmin = first ? 0 : -kVHist;
    const int nV = (n256 >> 4) - kmin;
    for (int q = tid; q < (nV >> 1); q += kThreads)
    {
      const int k = kmin + 2 * q;
      // U[4k-8 .. 4k+7] -> 8 dwords from (4k - 8 + kUHist)/2
      const uint32_t *up = lds + kUOff + ((4 * k - 8 + kUHist) >> 1);
      uint32_t u[8];
#pragma unroll
      for (int j = 0; j < 4; j++)
      {
        const uint2 t = *reinterpret_cast<const uint2 *>(up + 2 * j);
        u[2 * j] = t.x;
        u[2 * j + 1] = t.y;
      }
      int acc0 = 1 << 14, acc1 = 1 << 14;
#pragma unroll
      for (int j = 0; j < 6; j++)
      {
        acc0 = dot2(u[j], kRevD12.p[j], acc0);
        acc1 = dot2(u[j + 2], kRevD12.p[j], acc1);
      }
      const uint32_t w = ((uint32_t)q15_out(acc0) & 0xffffu) | ((uint32_t)q15_out(acc1) << 16);
      lds[kVOff + ((k + kVHist) >> 1)] = w;
    }
  }
  __syncthreads();

  // C4: PCM[p] = D(40,2)(V); two outputs per thread, one packed store
  {
    const int nP = n256 >> 5;
    uint32_t *pcm32 = reinterpret_cast<uint32_t *>(P.pcm + X.ounit * (size_t)nP);
    for (int q = tid; q < (nP >> 1); q += kThreads)
    {
      const int p = 2 * q;
      // V[2p-38 .. 2p+3] -> 21 dwords from (2p - 38 + kVHist)/2 = p
      const uint32_t *vq = lds + kVOff + p;
      int acc0 = 1 << 14, acc1 = 1 << 14;
      uint32_t prev = vq[0];
#pragma unroll
      for (int j = 0; j < 20; j++)
      {
        const uint32_t next = vq[j + 1];
        acc0 = dot2(prev, kRevD40.p[j], acc0);
        acc1 = dot2(next, kRevD40.p[j], acc1);
        prev = next;
      }
      pcm32[q] = ((uint32_t)q15_out(acc0) & 0xffffu) | ((uint32_t)q15_out(acc1) << 16);
    }
  }

  // carried histories for the next call
  if (last)
  {
    if (tid < kWbS / 2)
    {
      reinterpret_cast<uint32_t *>(so->wb_s)[tid] = lds[((n256 + kHist - kWbS) >> 1) + tid];
    }
    if (tid < kWbU)
    {
      so->wb_u[tid] = (int16_t)U16[kUHist + (n256 >> 2) - kWbU + tid];
    }
    if (tid < kWbV)
    {
      so->wb_v[tid] = (int16_t)V16[kVHist + (n256 >> 4) - kWbV + tid];
    }
  }
}

// =============================================================================
//  epilogue: squelch tracker over the batch, verification of both speculations,
//  n_pcm / allowed outputs.  One thread per channel.
// =============================================================================
__global__ void k_rx_epilogue(const EpilogueParams E)
{
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= E.n_channels)
  {
    return;
  }
  const int mode = E.cfg[c].mode;
  bool prev = E.state[c].tracking != 0;
  uint32_t gate_viol = 0, spec_viol = 0;
  for (uint32_t b = 0; b < E.n_blocks; b++)
  {
    const size_t unit = (size_t)c * E.n_blocks + b;
    const size_t ounit = (size_t)c * E.out_blocks + E.out_b0 + b;
    const bool present = E.present[unit] != 0;
    const bool allowed = present || prev;                 // Squelch.cc:227-273
    prev = present;                                       // SignalTracker.cc:104-146
    const bool demod = allowed && mode != 0;
    if (E.allowed != nullptr)
    {
      E.allowed[ounit] = allowed ? 1 : 0;
    }
    if (E.n_pcm != nullptr)
    {
      E.n_pcm[ounit] = demod ? E.n_pcm_per_block : 0u;
    }
    if (E.n_blocks > 1 && mode != 0 && !allowed)
    {
      gate_viol++;                                        // the batch assumed every gate open
    }
    if (b > 0 && mode == 3)
    {
      const uint32_t a = __builtin_bit_cast(uint32_t, E.chk_spec[unit]);
      const uint32_t p = __builtin_bit_cast(uint32_t, E.chk_pub[unit - 1]);
      if (a != p)
      {
        spec_viol++;
      }
    }
  }
  if (gate_viol)
  {
    atomicAdd(&E.counters[kCntGate], gate_viol);
  }
  if (spec_viol)
  {
    atomicAdd(&E.counters[kCntSpec], spec_viol);
  }
}

// commit the pending per-channel state when the whole call verified clean
__global__ void k_rx_commit(const EpilogueParams E)
{
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  const bool clean = (E.counters[kCntGate] | E.counters[kCntSpec]) == 0u;
  if (c == 0 && clean)
  {
    E.counters[kCntCommit] = 1u;
  }
  if (c >= E.n_channels || !clean)
  {
    return;
  }
  ChanState *dst = E.state + c;
  const ChanState *src = E.state_out + c;
  const int mode = E.cfg[c].mode;
  // tracker over the batch; was the last block demodulated?
  bool prev = dst->tracking != 0;
  bool allowed = false;
  for (uint32_t b = 0; b < E.n_blocks; b++)
  {
    const bool present = E.present[(size_t)c * E.n_blocks + b] != 0;
    allowed = present || prev;
    prev = present;
  }
  dst->tracking = prev ? 1u : 0u;
  for (int i = 0; i < 16; i++)
  {
    dst->fe_tail[i] = src->fe_tail[i];
  }
  if (!allowed)
  {
    return;                                              // demodulator state frozen
  }
  if (mode == 3)
  {
    dst->wb_theta = src->wb_theta;
    dst->wb_p = src->wb_p;
    dst->wb_y = src->wb_y;
    for (int i = 0; i < kWbS; i++) dst->wb_s[i] = src->wb_s[i];
    for (int i = 0; i < kWbU; i++) dst->wb_u[i] = src->wb_u[i];
    for (int i = 0; i < kWbV; i++) dst->wb_v[i] = src->wb_v[i];
  }
}

// explicit instantiations used by the host side
template __global__ void k_rx_wbfm<0>(const RxParams);
template __global__ void k_rx_wbfm<3>(const RxParams);

} // namespace hrfd
