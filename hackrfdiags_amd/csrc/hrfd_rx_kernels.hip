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
#include <type_traits>
#include <utility>
#include "hrfd_tables.h"
#ifndef HRFD_IIR_U
#define HRFD_IIR_U 10   /* divides kTile and the warm-up: no scalar tail */
#endif
#ifdef HRFD_ABLATE
#define HRFD_ABLATE_EARLY HRFD_ABLATE
#else
#define HRFD_ABLATE_EARLY 0
#endif


namespace hrfd {

typedef short s2 __attribute__((ext_vector_type(2)));    // one (I,Q) pair or two taps

__device__ __forceinline__ s2 as_s2(uint32_t u) { return __builtin_bit_cast(s2, u); }
__device__ __forceinline__ uint32_t as_u32(s2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// ---- cross-lane neighbours without LDS ------------------------------------------
// prev(v): lane i <- v[i-1]; lane 0 <- lane 0 of `carry` (v_mov_b32_dpp wave_shr:1,
// lanes without a source keep the tied `old` operand).
__device__ __forceinline__ uint32_t shr1(uint32_t v, uint32_t carry)
{
  return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)v, 0x138, 0xf, 0xf, false);
}
// lane 0 <- v[63] (v_mov_b32_dpp wave_ror:1): the carry for the NEXT chunk, kept
// in a VGPR so that no readlane / scalar round trip is needed.
__device__ __forceinline__ uint32_t ror1(uint32_t v)
{
  int dontcare;                                          // every lane has a source: `old` is never read
  asm volatile("" : "=v"(dontcare));
  return (uint32_t)__builtin_amdgcn_update_dpp(dontcare, (int)v, 0x13C, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t lane63(uint32_t v)
{
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// ---- half-band /2 stages on (I,Q) pairs, offset-binary domain ----------------
// Reference: y = (16384 + h0*(a+c) + 16384*b) >> 15 with h0 = 8192 + d
// (Decimator_int16.cc:176-249 with the 3-tap tables of IqDataProcessor.cc:8-27).
// With a' = a+128 etc. and T = a'+c':  4*(y+128) + rem = T + 2b' + k(T), where
//   k(T) = floor((d*(T-256) + 16384) / 8192)
//        = 1 + (T >> 8)                     h0 = 8206 (d = 14)   [stage 1: on bytes, hb1_bytes below]
//        = (57 T + 1792) >> 13              h0 = 8249 (d = 57)
//        = ((29 T + 2816) >> 10) - 8        h0 = 8424 (d = 232)
// I sits in bits 0..15 and Q in bits 16..31 of one register.  Every field stays
// in 0..2053, so plain 32-bit adds and (shift, mask) pairs act on both fields at
// once; those are 32-bit-encoded VOP2 instructions, which gfx950 issues at about
// twice the rate of the 64-bit-encoded packed-math ones (tools/ubench/valu_rate.hip).
// Stages 2 and 3 run that way (only their multiplies are v_pk_mad_u16).  The last stage adds 4*256 so that
// no field goes negative; the (int8_t) narrowing of IqDataProcessor.cc:458,489
// keeps the low byte only, which that does not touch.
// Checked exhaustively against the direct form in tests/test_abi_and_tables.py.
// Round 4: the kernel is bound by the NUMBER of vector instructions it issues (SQ counters: one VALU instruction per
// SIMD and ~4 cycles, whatever its encoding -- profiles/r4_sq_counters.txt), so a (shift, mask) pair on both fields is
// ONE packed shift (v_pk_lshrrev_b16: the fields cannot leak into each other, nothing to mask), stage 3's constant rides
// on the three-input add that forms T, and the doubled centre tap of stage 3 on a shift-and-add.
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
// (the packed forms below wrap modulo 2^16 per field on purpose: UNSIGNED fields, for which that is defined -- signed
//  ext-vector arithmetic would be undefined on overflow; the same v_pk_mad_u16 / v_pk_sub_u16 are emitted)
__device__ __forceinline__ us2 as_us2(uint32_t u) { return __builtin_bit_cast(us2, u); }
__device__ __forceinline__ uint32_t as_u32u(us2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t pk_shr(uint32_t v, unsigned short k)
{
  return __builtin_bit_cast(uint32_t, (us2)(__builtin_bit_cast(us2, v) >> k));
}
// b2 = 2 b' (the centre tap, doubled)
__device__ __forceinline__ uint32_t hb2_sum(uint32_t a, uint32_t b2, uint32_t c)
{
  const uint32_t t = a + c;
  const uint32_t k = pk_shr(as_u32u(as_us2(t) * (unsigned short)57 + (unsigned short)1792), 13);    // 0..3 per field
  return t + b2 + k;
}
// b = b' (the centre tap as the previous stage's output, NOT doubled).  T' = T + (-8 + 4*256) per field feeds the
// multiply as well: 29 T' + 38888 = 29 T + 2816 + 65536, and the packed multiply-add keeps 16 bits.
__device__ __forceinline__ uint32_t hb3_sum(uint32_t a, uint32_t b, uint32_t c)
{
  const uint32_t t = a + c + 0x03f803f8u;
  const uint32_t k = pk_shr(as_u32u(as_us2(t) * (unsigned short)29 + (unsigned short)38888), 10);   // 2..17 per field (modulo 2^16: unsigned fields, defined wrap-around)
  return ((b << 1) + t) + k;
}
// a stage-2 sum (<= 1023 per field) as the next stage's tap y'
__device__ __forceinline__ uint32_t form_ac(uint32_t s) { return pk_shr(s, 2); }

// Stage 1 (h0 = 8206) works on BYTES, four at a time, with the pixel-average instruction
// v_lerp_u8 (per byte (x + y + r) >> 1, r = bit 0 of the matching byte of the third operand).
// With m = (a' + c') >> 1 and e = (a' + c') & 1 the sum above is 2(m + b') + e + 1 + [m >= 128],
// so  y' = (m + b' + (e | m>>7)) >> 1 = lerp(lerp(a', c', 0), b', (a' ^ c') | (m >> 7)):
// five instructions for two (I,Q) outputs, straight from the interleaved input bytes -- no
// unpacking to 16-bit fields before the first stage (tests: all 2^24 byte triples).
__device__ __forceinline__ uint32_t hb1_bytes(uint32_t a, uint32_t b, uint32_t c)
{
  const uint32_t m = __builtin_amdgcn_lerp(a, c, 0u);
  const uint32_t r = (a ^ c) | (m >> 7);                 // only bit 0 of each byte is looked at
  return __builtin_amdgcn_lerp(m, b, r);
}

// Carry between consecutive 1 KiB chunks of one wave: lane 0 of each register
// holds the value of the previous chunk's last lane.
struct FeCarry
{
  uint32_t x7;    // last input dword (offset binary): its upper two bytes are the last (I,Q) pair
  uint32_t y13;   // last stage-1 output pair
  uint32_t y21;   // last stage-2 output pair
};

// 16 raw bytes = 8 IQ samples -> one 256 kS/s sample (offset-binary, low byte
// of each half = value + 128).  IqDataProcessor::reduceSampleRate, :429-500.
__device__ __forceinline__ uint32_t frontend(const uint4 raw, FeCarry &c)
{
  // dword k holds samples x[2k], x[2k+1] as bytes (I, Q, I, Q)
  const uint32_t r0 = raw.x ^ 0x80808080u, r1 = raw.y ^ 0x80808080u;
  const uint32_t r2 = raw.z ^ 0x80808080u, r3 = raw.w ^ 0x80808080u;
  const uint32_t rm1 = shr1(r3, c.x7);                   // the dword in front of this lane's 16 bytes
  c.x7 = ror1(r3);
  // stage 1, outputs (y1[0], y1[1]) and (y1[2], y1[3]) as byte quadruples:
  // outer taps x[2m-1], x[2m+1] (odd samples), centre x[2m] (even samples)
  const uint32_t a01 = __builtin_amdgcn_perm(r0, rm1, 0x07060302u);   // x[-1], x[1]
  const uint32_t b01 = __builtin_amdgcn_perm(r1, r0, 0x05040100u);    // x[0],  x[2]
  const uint32_t c01 = __builtin_amdgcn_perm(r1, r0, 0x07060302u);    // x[1],  x[3]
  const uint32_t a23 = __builtin_amdgcn_perm(r2, r1, 0x07060302u);    // x[3],  x[5]
  const uint32_t b23 = __builtin_amdgcn_perm(r3, r2, 0x05040100u);    // x[4],  x[6]
  const uint32_t c23 = __builtin_amdgcn_perm(r3, r2, 0x07060302u);    // x[5],  x[7]
  const uint32_t y01 = hb1_bytes(a01, b01, c01);
  const uint32_t y23 = hb1_bytes(a23, b23, c23);
  // to 16-bit (I,Q) fields for the other two stages
  const uint32_t y10 = __builtin_amdgcn_perm(0u, y01, 0x0c010c00u);
  const uint32_t y11 = __builtin_amdgcn_perm(0u, y01, 0x0c030c02u);
  const uint32_t y12 = __builtin_amdgcn_perm(0u, y23, 0x0c010c00u);
  const uint32_t y13 = __builtin_amdgcn_perm(0u, y23, 0x0c030c02u);

  const uint32_t y1m1 = shr1(y13, c.y13);
  c.y13 = ror1(y13);
  const uint32_t y20b = form_ac(hb2_sum(y1m1, y10 << 1, y11));
  const uint32_t y21 = form_ac(hb2_sum(y11, y12 << 1, y13));

  const uint32_t y2m1 = shr1(y21, c.y21);
  c.y21 = ror1(y21);
  return hb3_sum(y2m1, y20b, y21) >> 2;                  // low byte of each half significant
}

// The carry that a chunk boundary needs is a function of the 16 bytes before
// it only (x1..x7 of that slot): recompute it from those bytes.
__device__ __forceinline__ FeCarry carry_from_16(const uint4 raw)
{
  const uint32_t r0 = raw.x ^ 0x80808080u, r1 = raw.y ^ 0x80808080u;
  const uint32_t r2 = raw.z ^ 0x80808080u, r3 = raw.w ^ 0x80808080u;
  const uint32_t b01 = __builtin_amdgcn_perm(r1, r0, 0x05040100u);
  const uint32_t c01 = __builtin_amdgcn_perm(r1, r0, 0x07060302u);
  const uint32_t a23 = __builtin_amdgcn_perm(r2, r1, 0x07060302u);
  const uint32_t b23 = __builtin_amdgcn_perm(r3, r2, 0x05040100u);
  const uint32_t c23 = __builtin_amdgcn_perm(r3, r2, 0x07060302u);
  // y1[1] needs x[1..3] only; y1[0] is not needed (its outer tap x[-1] lies before the 16 bytes)
  const uint32_t a01 = __builtin_amdgcn_perm(r0, r0, 0x07060302u);    // (x[1], x[1]): pair 1 is what counts
  const uint32_t y01 = hb1_bytes(a01, b01, c01);
  const uint32_t y23 = hb1_bytes(a23, b23, c23);
  const uint32_t y11 = __builtin_amdgcn_perm(0u, y01, 0x0c030c02u);
  const uint32_t y12 = __builtin_amdgcn_perm(0u, y23, 0x0c010c00u);
  const uint32_t y13 = __builtin_amdgcn_perm(0u, y23, 0x0c030c02u);
  FeCarry c;
  c.x7 = r3;
  c.y13 = y13;
  c.y21 = form_ac(hb2_sum(y11, y12 << 1, y13));
  return c;
}

// ---- Fs/4 rotation + table index + squelch magnitude -------------------------
// y3 holds (I+128, Q+128) (low bytes significant: the (int8_t) narrowing of
// IqDataProcessor.cc:458,489).  upconvertByFsOver4 (:771-815) multiplies by
// {1, j, -1, -j}: rot 0 (I,Q), 1 (-Q,I), 2 (-I,-Q), 3 (Q,-I), int8 negation
// wrapping.  In index form (value+128) negation is (256 - idx) & 255 =
// ((idx ^ 0xff) + 1) & 0xff, done on both halves at once with per-lane constants.
struct MixConst
{
  uint32_t swap;   // all ones when the lane's rotation swaps I and Q
  uint32_t xorm;   // 0xff in the halves that are negated
  uint32_t addc;   // 1 in the halves that are negated
};

__device__ __forceinline__ MixConst mix_const(int rot)
{
  MixConst m;
  m.swap = (rot & 1) ? 0xffffffffu : 0u;
  const uint32_t na = (rot == 1 || rot == 2) ? 1u : 0u;    // new I negated
  const uint32_t nb = (rot == 2 || rot == 3) ? 1u : 0u;    // new Q negated
  m.xorm = (na ? 0x000000ffu : 0u) | (nb ? 0x00ff0000u : 0u);
  m.addc = na | (nb << 16);
  return m;
}

// returns (q_idx << 16) | i_idx -- (uint8)(Q+128), (uint8)(I+128) after the mix
__device__ __forceinline__ uint32_t mix_fs4(uint32_t y3, const MixConst mc)
{
  const uint32_t sw = __builtin_amdgcn_alignbit(y3, y3, 16);      // halves exchanged
  const uint32_t ab = (mc.swap != 0u) ? sw : y3;
  return ((ab ^ mc.xorm) + mc.addc) & 0x00ff00ffu;
}

// max(|I|,|Q|) + (min(|I|,|Q|) >> 1) of the int8 values (SignalDetector.cc:226-241);
// |-128| = 128 as in the reference's uint8.  Rotation invariant (the Fs/4 mixer only
// swaps and negates, and -(-128) wraps to -128), so it is taken from the rotated pair,
// whose |i|, |q| the arithmetic atan2 needs anyway.
__device__ __forceinline__ uint32_t magnitude(uint32_t y3)
{
  const s2 d = as_s2(y3 & 0x00ff00ffu) - as_s2(0x00800080u);
  const s2 nd = as_s2(0u) - d;
  const s2 ad = __builtin_elementwise_max(d, nd);
  const uint32_t ai = (uint32_t)(uint16_t)ad.x, aq = (uint32_t)(uint16_t)ad.y;
  return max(ai, aq) + (min(ai, aq) >> 1);
}

// ---- arithmetic atan2 ---------------------------------------------------------
// The reference looks theta up in a 256 x 256 float table built with libm
// (WbFmDemodulator.cc:137-148).  A divergent table gather costs the texture
// path ~250 cycles per wave (measured, tools/ubench/stream_gather_asm.hip),
// which made it the bound of the whole kernel.  Instead theta is computed:
//   a = max(|i|,|q|), b = min(|i|,|q|), r = b/a, phi = r + r^3 P(r^2)  (~1 ulp)
//   theta0 = phi | pi/2 - phi | pi - phi | pi - (pi/2 - phi)   by octant,
// then made bit-identical to the table with a 2-bit (signed) correction per
// (a, b, octant class) -- 8385 bytes, resident in LDS -- and the sign of q.
// The correction bytes are derived on the device at hrfd_rx_create() from the
// very table they replace, with this very function (k_build_atan_corr), so the
// result is the table's by construction; if a correction does not fit 2 bits the
// library keeps the gather kernel.
constexpr float kAtanC[8] = {-0.333329797f, 0.199902073f, -0.141844273f, 0.105678506f, -0.0735382065f, 0.0409709625f, -0.0150405606f, 0.00259946357f};
constexpr float kPiF = 3.14159274f, kPi2F = 1.57079637f;

// approximate theta for q >= 0, and the bit offset of its 2-bit correction
struct AtanApprox
{
  float theta0;
  uint32_t shift;
};

__device__ __forceinline__ AtanApprox atan2_approx(uint32_t a, uint32_t b, bool swap, bool negi, float inv_a)
{
#if (HRFD_ABLATE_EARLY & 512)
  // TIMING EXPERIMENT ONLY: phi straight from the LDS word (what a first-octant float table would cost)
  {
    float v = swap ? (kPi2F - inv_a) : inv_a;
    v = negi ? (kPiF - v) : v;
    AtanApprox o;
    o.theta0 = v;
    o.shift = (swap ? 2u : 0u) | (negi ? 4u : 0u);
    return o;
  }
#endif
  const float bf = (float)b, af = (float)a;
  const float r0 = bf * inv_a;
  const float e = __builtin_fmaf(-af, r0, bf);
  const float r = __builtin_fmaf(e, inv_a, r0);         // b / a, correctly rounded
  const float s = r * r;
  float p = kAtanC[7];
#pragma unroll
  for (int k = 6; k >= 0; k--)
  {
    p = __builtin_fmaf(p, s, kAtanC[k]);
  }
  const float t = r * s;
  const float phi = __builtin_fmaf(t, p, r);
  float v = swap ? (kPi2F - phi) : phi;
  v = negi ? (kPiF - v) : v;
  AtanApprox o;
  o.theta0 = v;
  o.shift = (swap ? 2u : 0u) | (negi ? 4u : 0u);
  return o;
}

// mixed = (q_idx << 16) | i_idx (offset binary); corr / inv are the LDS copies
__device__ __forceinline__ float theta_arith(uint32_t mixed, const uint8_t *corr, const float *inv)
{
  const s2 d = as_s2(mixed) - as_s2(0x00800080u);
  const s2 nd = as_s2(0u) - d;
  const s2 ad = __builtin_elementwise_max(d, nd);
  const uint32_t ai = (uint32_t)(uint16_t)ad.x, aq = (uint32_t)(uint16_t)ad.y;
  const bool swap = aq > ai;
  const uint32_t a = max(ai, aq), b = min(ai, aq);
  const uint32_t tri = ((a * a + a) >> 1) + b;
#if (HRFD_ABLATE_EARLY & 64)
  const uint32_t code8 = 0x55u + (tri & 0u);                // TIMING EXPERIMENT ONLY: no LDS reads
  const float inv_a = 0.0078125f;
#else
  const uint32_t code8 = corr[tri];
#if (HRFD_ABLATE_EARLY & 512)
  const float inv_a = inv[tri & 127u];
#else
  const float inv_a = inv[a];
#endif
#endif
  const bool negi = (mixed & 0x00000080u) == 0u;
  const AtanApprox ap = atan2_approx(a, b, swap, negi, inv_a);
  // 2-bit two's complement field: (exact - approx) in ulps, -2..1
  const int32_t fix = __builtin_amdgcn_sbfe((int32_t)code8, ap.shift, 2u);
  const uint32_t bits = f2u(ap.theta0) + (uint32_t)fix;
  // q < 0 (bit 23 of `mixed` clear): atan2(-q, i) = -atan2(q, i); theta0 >= 0, so set the sign:
  // bits | (~(mixed << 8) & 0x80000000) as one three-input bit operation
  return u2f(__builtin_amdgcn_bitop3_b32(bits, mixed << 8, 0x80000000u, 0xF2));
}

// ---- atan2 from a first-octant table -------------------------------------------
// k_rx_wbfm_flow has the LDS for one more table: T0[a(a+1)/2 + b] = (float)atan2((double)b, (double)a),
// 0 <= b <= a <= 128, built by the host's libm exactly like the reference's own table
// (WbFmDemodulator.cc:137-148) -- for |q| <= |i|, i > 0 it IS the reference's entry.  The other
// octants are pi/2 - t, pi - t, pi - (pi/2 - t) in float plus the same kind of 2-bit correction
// (-1, 0 or +1 ulp here), derived on the device from the reference table with this very function
// (k_build_atan_corr<true>).  33 vector instructions of theta_arith become two subtractions.
__device__ __forceinline__ AtanApprox atan2_from_t0(float t0, bool swap, bool negi)
{
  float v = swap ? (kPi2F - t0) : t0;
  v = negi ? (kPiF - v) : v;
  AtanApprox o;
  o.theta0 = v;
  o.shift = (swap ? 2u : 0u) | (negi ? 4u : 0u);
  return o;
}

__device__ __forceinline__ float theta_tab(uint32_t mixed, const uint8_t *corr, const float *t0tab)
{
#if (HRFD_ABLATE_EARLY & 8192)
  // TIMING EXPERIMENT ONLY (wrong values): the instruction shape of a first-QUADRANT table whose words carry the 2-bit
  // correction of the i < 0 half in their two free top bits
  {
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const s2 d = as_s2(mixed) - as_s2(0x00800080u);
    const s2 nd = as_s2(0u) - d;
    const s2 ad = __builtin_elementwise_max(d, nd);
    const uint32_t off = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, ad), __builtin_bit_cast(us2, (65u * 4u) << 16 | 4u), 0u, false);
    const uint32_t w = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(t0tab) + off);
    const uint32_t t = w & 0x3fffffffu;
    const int32_t fix = (int32_t)w >> 30;
    const uint32_t pv = f2u(kPiF - u2f(t)) + (uint32_t)fix;
    const uint32_t m = (uint32_t)((int32_t)(mixed << 24) >> 31);   // all ones: i >= 0
    const uint32_t sel = __builtin_amdgcn_bitop3_b32(t, pv, m, 0xE4);   // m ? t : pv
    return u2f(__builtin_amdgcn_bitop3_b32(sel, mixed << 8, 0x80000000u, 0xF2));
  }
#endif
  const s2 d = as_s2(mixed) - as_s2(0x00800080u);
  const s2 nd = as_s2(0u) - d;
  const s2 ad = __builtin_elementwise_max(d, nd);
  const uint32_t ai = (uint32_t)(uint16_t)ad.x, aq = (uint32_t)(uint16_t)ad.y;
  const bool swap = aq > ai;
  const uint32_t a = max(ai, aq), b = min(ai, aq);
  const uint32_t tri = ((a * a + a) >> 1) + b;
  const uint32_t code8 = corr[tri];
  const float t0 = t0tab[tri];
  const bool negi = (mixed & 0x00000080u) == 0u;
  const AtanApprox ap = atan2_from_t0(t0, swap, negi);
  const int32_t fix = __builtin_amdgcn_sbfe((int32_t)code8, ap.shift, 2u);
  const uint32_t bits = f2u(ap.theta0) + (uint32_t)fix;
  return u2f(__builtin_amdgcn_bitop3_b32(bits, mixed << 8, 0x80000000u, 0xF2));
}

// ---- atan2 from a first-QUADRANT table (round 5: the re-split flow kernel) -------------------------------------
// TQ[|q| * 129 + |i|], 0 <= |i|, |q| <= 128: for i >= 0, q >= 0 the word IS the reference's entry
// (WbFmDemodulator.cc:137-148: (float)atan2((double)q, (double)i), host libm).  Every entry is below 2.0, so bits 31
// and 30 of its float are zero: they carry the signed 2-bit correction (in ulps, -2 .. 1) that makes
// bits(pi_f - t) + fix the reference's entry for i < 0 (built and PROVEN to fit by the host at hrfd_rx_create from the
// very table it replaces: build_atan_quadrant in hrfd_api.hip).  q < 0 negates (the reference table is odd in q:
// checked there too).  One LDS word per sample, no octant logic: |i|, |q| come four bytes at a time.
// x = the ring word of two samples as SIGNED bytes (i0, q0, i1, q1) -> |.| of the four bytes (|-128| = 128)
__device__ __forceinline__ uint32_t abs4_s8(uint32_t x)
{
  const uint32_t s = (x >> 7) & 0x01010101u;            // 1 in the bytes that are negative
  uint32_t s8 = s << 8;
  asm("" : "+v"(s8));                                    // (left to itself the compiler makes s * 255 of the next line: v_mul_lo_u32, quarter rate)
  const uint32_t m = s8 - s;                             // 0xff there
  return (x ^ m) + s;                                    // ~b + 1 <= 128: no carry leaves a byte
}

// theta of sample k (0 or 1) of the word, in two halves so that a caller can put many lookups in flight before it
// uses the first: the table index (x signed bytes, a = abs4_s8(x)) ...
template <int K>
__device__ __forceinline__ uint32_t theta_quad_index(uint32_t a)
{
  const uint32_t coef = (K == 0) ? 0x00008101u : 0x81010000u;       // |i| + 129 |q| of sample K
  return __builtin_amdgcn_udot4(a, coef, 0u, false);
}
// ... and theta from the table's word w
template <int K>
__device__ __forceinline__ float theta_quad_word(uint32_t x, uint32_t w)
{
  const uint32_t t = w & 0x3fffffffu;
  const int32_t fix = (int32_t)w >> 30;
  const uint32_t pv = f2u(kPiF - u2f(t)) + (uint32_t)fix;
  // i < 0 as a mask of 32 bits (one v_bfe_i32) and the choice as a bitfield insert: two instructions where a test, a
  // compare and a select were three (and a wait state for vcc)
  uint32_t ineg = (uint32_t)__builtin_amdgcn_sbfe((int32_t)x, (K == 0) ? 7u : 23u, 1u);
  asm("" : "+v"(ineg));                                  // (seen through, the compiler turns it back into the three)
  const uint32_t mag = (pv & ineg) | (t & ~ineg);
  // sign of q: bit 15 (sample 0) / bit 31 (sample 1) of x
  const uint32_t sg = (K == 0) ? (x << 16) : x;
  return u2f(__builtin_amdgcn_bitop3_b32(mag, sg, 0x80000000u, 0xF8));   // mag | (sg & 0x80000000): index = 4a + 2b + c
}
template <int K>
__device__ __forceinline__ float theta_quad(uint32_t x, uint32_t a, const uint32_t *tq)
{
  return theta_quad_word<K>(x, tq[theta_quad_index<K>(a)]);
}

// deltaTheta wrap (WbFmDemodulator.cc:417-425).  The reference compares the
// float against the double M_PI: (double)d > M_PI  <=>  d >= 0x1.921fb6p+1f,
// and subtracts 2*M_PI in double before rounding back to float.  |d| <= 2*pi,
// so one correction suffices.  The double subtraction is replaced by two float
// ones, (d -+ C_HI) -+ C_LO with C_HI + C_LO = 2*M_PI to float precision: the first
// is exact (Sterbenz), the second rounds once, and tools/proofs/wrap_float.c shows
// over all 4.0e8 wrapping pairs of table thetas that it rounds to the same float.
__device__ __forceinline__ float wrap_pi(float d)
{
  const float pi_up = 3.14159274101257324e+00f;          // smallest float > M_PI
  const uint32_t c_hi = 0x40c90fdbu;                     // (float)(2*M_PI)
  const uint32_t c_lo = 0xb43bbd2eu;                     // (float)(2*M_PI - C_HI) = -0x1.777a5cp-23
  const uint32_t sg = f2u(d) & 0x80000000u;
  const float u = d - u2f(c_hi | sg);
  const float w = u - u2f(c_lo ^ sg);
  return (fabsf(d) >= pi_up) ? w : d;
}

// The same wrap without compare and select: n = rint(d * CM) is 0 or +-1, with CM the float for which the flip
// falls exactly between the largest float below M_PI and the smallest above it; d - n*C_HI is exact, the second
// fma rounds once like the second subtraction of wrap_pi.  tools/proofs/wrap_rint.c checks EVERY float d with
// |d| <= 6.5 against the reference's double arithmetic (d = -0.0, which a difference of table thetas never is,
// comes out as +0.0).  Four instructions for seven.
__device__ __forceinline__ float wrap_pi_rint(float d)
{
  const float cm = u2f(0x3e22f984u);                     // ~ 1 / (2 pi)
  const float c_hi = u2f(0x40c90fdbu);                   // (float)(2*M_PI)
  const float c_lo = u2f(0xb43bbd2eu);                   // (float)(2*M_PI - C_HI)
  const float n = __builtin_rintf(d * cm);
  const float u = __builtin_fmaf(-n, c_hi, d);
  return __builtin_fmaf(-n, c_lo, u);
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

template <bool FIX = true>
__device__ __forceinline__ uint32_t pack_s16(float lo, float hi)
{
  if (FIX)
  {
    return ((uint32_t)f2i16(lo) & 0xffffu) | ((uint32_t)f2i16(hi) << 16);
  }
  // |y| < 2^31 is known: v_cvt_i32_f32 cannot saturate, the low 16 bits are x86's (one v_perm packs the two low halves)
  return __builtin_amdgcn_perm((uint32_t)(int)hi, (uint32_t)(int)lo, 0x05040100u);
}

// Have two runs of the de-emphasis recurrence become the same trajectory?
// Bitwise-equal values stay equal forever (same inputs from here on).  +0/-0
// compare equal: the next step erases the sign and (int16_t) maps both to 0.
// One more case is accepted: both values denormal or zero.  On silent input the
// rounded recurrence y <- 0.949*y has the fixed points k*2^-149, |k| <= 9, so a
// decayed tail and a from-zero run may never meet bit for bit; both convert to
// PCM 0, and their distance (< 2^-126) is far below half an ulp of any value
// that could reach the PCM later, where both runs round to the same float.
// NaN never matches (forces the exact path).
__device__ __forceinline__ bool same_trajectory(float a, float b)
{
  const float tiny = 1.17549435e-38f;                  // FLT_MIN
  return (a == b) || (fabsf(a) < tiny && fabsf(b) < tiny);
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
// the FIR demodulators' decimators (FmDemodulator.cc:17-51, AmDemodulator.cc:14-62 == SsbDemodulator.cc:14-62)
__constant__ constexpr RevTaps<N_FM_TUNER_D32> kRevTuner(Q_FM_TUNER_D32);
__constant__ constexpr RevTaps<N_AM_D1> kRevAmD1(Q_AM_D1);
__constant__ constexpr RevTaps<N_AM_D2> kRevAmD2(Q_AM_D2);
__constant__ constexpr RevTaps<N_AM_D3> kRevAmD3(Q_AM_D3);
// A Q15 stage fed with OFFSET-BINARY samples u = x + 128 (what the ring of k_rx_wbfm_flow's FIR modes holds): the
// sum over h (u - 128) is the sum over h u minus 128 sum h, so the rounding constant absorbs the offset (int32
// wrap-around arithmetic: identical bits)
template <int N>
constexpr int q15_bias_init(const int16_t (&h)[N])
{
  long long sum = 0;
  for (int i = 0; i < N; i++)
  {
    sum += h[i];
  }
  return (int)((1 << 14) - 128 * sum);
}

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
constexpr int kPairsPerThread = ((kMaxN256 + kHist) / 2 + kThreads - 1) / kThreads;

// ---- the two audio-rate integer stages shared by WBFM and FM ------------------
// U (64 kS/s, int16) lives at dword kUOff with kUHist samples of history in
// front, V (16 kS/s) at kVOff with kVHist; both tables are the reference's
// postDemodDecimator2 / audioDecimator (WbFmDemodulator.cc:28-86 ==
// FmDemodulator.cc:53-111).
constexpr int kUOff = (kSDwords + 31) / 32 * 32;               // dword offset of U (16-B aligned): 8576
constexpr int kUHist = 160;
constexpr int kVOff = kUOff + (kMaxN256 / 4 + kUHist) / 2;    // 10704
constexpr int kVHist = 38;
static_assert(kVOff + (kMaxN256 / 16 + kVHist) / 2 + 1 <= kMaxNV, "LDS map");
static_assert(kSDwords <= kUOff && (kUOff % 4) == 0, "LDS map");

// V[k] = D(12,4)(U) for k in [kmin, n16); kmin even; two outputs per thread
// (ubase / vbase: dword pointers to U[-kUHist] and V[-kVHist]; nthreads threads take part)
__device__ __forceinline__ void stage_d12(const uint32_t *ubase, uint32_t *vbase, const int kmin, const int n16,
                                          const int tid, const int nthreads = kThreads)
{
  const int nV = n16 - kmin;
  for (int q = tid; q < (nV >> 1); q += nthreads)
  {
    const int k = kmin + 2 * q;
    // U[4k-8 .. 4k+7] -> 8 dwords from (4k - 8 + kUHist)/2
    const uint32_t *up = ubase + ((4 * k - 8 + kUHist) >> 1);
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
    vbase[(k + kVHist) >> 1] = w;
  }
}

// PCM[p] = D(40,2)(V) for p in [0, nP); two outputs per thread, one packed store
__device__ __forceinline__ void stage_d40(const uint32_t *vbase, const int nP, uint32_t *pcm32, const int tid,
                                          const int nthreads = kThreads)
{
  for (int q = tid; q < (nP >> 1); q += nthreads)
  {
    const int p = 2 * q;
    // V[2p-38 .. 2p+3] -> 21 dwords from (2p - 38 + kVHist)/2 = p
    const uint32_t *vq = vbase + p;
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

// One lane's run of y[n] = v[n] - a1*y[n-1] over `count` consecutive samples at
// `in` (IirFilter.cc:161-176: r = a1*y; y = v - r -- two rounded operations).
// A lone wave issues one instruction every ~5 cycles, so this loop is bound by
// its instruction count: per 16 steps it spends 8 ds_read2_b32 (issued one group
// ahead of the dependent chain), 32 multiply/subtract and, when STORE, 8
// ds_write2_b32; the two register groups alternate roles, no copies.
// Steps k < kskip leave y untouched (select, no branch; SKIP variant only).
template <bool STORE, bool SKIP>
__device__ __forceinline__ float iir_run(const uint32_t *in, uint32_t *out, const int count,
                                         const int kskip, float y)
{
  const float a1 = DEEMPH_A1;
  constexpr int U = HRFD_IIR_U;                        // steps per prefetched group
  static_assert(U % 2 == 0, "groups are read as 64-bit pairs");
  auto step = [&](float vv, int k) {
    const float r = a1 * y;
    const float yn = vv - r;
    y = (!SKIP || k >= kskip) ? yn : y;
  };
  // Every lane's range starts on an even dword (rx_launch keeps the tile length, the warm-up
  // and the history even) and groups start at multiples of U: 64-bit LDS accesses, half as
  // many LDS instructions between the dependent multiply-subtract pairs.
  auto load_group = [&](float (&g)[U], int at) {
    const uint2 *p2 = reinterpret_cast<const uint2 *>(in + at);
#pragma unroll
    for (int j = 0; j < U / 2; j++)
    {
      const uint2 w = p2[j];
      g[2 * j] = u2f(w.x);
      g[2 * j + 1] = u2f(w.y);
    }
  };
  auto run_group = [&](const float (&g)[U], int at) {
    uint2 *o2 = reinterpret_cast<uint2 *>(out + at);
#pragma unroll
    for (int j = 0; j < U / 2; j++)
    {
      step(g[2 * j], at + 2 * j);
      const float y0 = y;
      step(g[2 * j + 1], at + 2 * j + 1);
      if (STORE)
      {
        o2[j] = make_uint2(f2u(y0), f2u(y));
      }
    }
  };
  int k = 0;
  float ga[U], gb[U];
  if (count >= U)
  {
    load_group(ga, 0);
  }
  // two groups per iteration: (ga: steps k..k+U-1, gb: k+U..k+2U-1)
  for (; k + 3 * U <= count; k += 2 * U)
  {
    load_group(gb, k + U);
    run_group(ga, k);
    load_group(ga, k + 2 * U);
    run_group(gb, k + U);
  }
  // here ga holds steps k..k+U-1 when k + U <= count
  if (k + U <= count)
  {
    run_group(ga, k);
    k += U;
  }
  for (; k < count; k++)
  {
    step(u2f(in[k]), k);
    if (STORE)
    {
      out[k] = f2u(y);
    }
  }
  return y;
}

// Everything phase A needs to (re)produce a range of the 256 kS/s stream.
struct StreamCtx
{
  const RxParams *P;
  __amdgpu_buffer_rsrc_t rsrc;   // the channel's raw input (all blocks of the call)
  uint32_t blk_off;              // byte offset of this block inside it
  const ChanState *st;
  uint32_t *lds;
  size_t ounit;
  float kgain;
  int hal, vstart, n256;
  int lane;
  int qoff;                      // FIR modes: int16 index of the Q rail inside lds (I rail at 0)
  const uint8_t *atc;            // arithmetic atan2: LDS copies of the correction bytes and of 1/a
  const float *ati;              //   (quad_piece<2>: of the first-octant table T0)
  uint4 tab;                     // this thread's 16 bytes of them, loaded at kernel entry (in flight)
  bool publish;                  // ... and still to be published to LDS by this produce_stream call
  bool first;
};

// cache policy bits of the streaming loads (bit 0 sc0, bit 1 nt, bit 4 sc1)
#ifndef HRFD_STREAM_AUX
#define HRFD_STREAM_AUX 0
#endif

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// One 1 KiB chunk of raw input (64 lanes x 16 bytes) as a buffer load: the lane
// offset is a constant VGPR, the chunk offset a scalar, so a load costs no
// vector ALU work; reads past the end of the channel's input return zeros.
template <bool S256>
__device__ __forceinline__ uint4 load_chunk(const StreamCtx &X, int chunk, int limit)
{
  // prefetches past the end of the run (chunk >= limit) are pointed outside the buffer: they
  // return zeros without touching memory, and the pipeline needs no branch
  const bool live = chunk < limit;
  if (S256)
  {
    // inner demodulator API: the stream is already at 256 kS/s, one (I,Q) byte pair per lane
    const uint32_t soff = live ? X.blk_off + (uint32_t)((X.vstart + 64 * chunk) * 2) : 0xfffff000u;
    const uint32_t w = __builtin_amdgcn_raw_buffer_load_b16(X.rsrc, X.lane * 2, soff, 0);
    return make_uint4(w, 0u, 0u, 0u);
  }
#if (HRFD_ABLATE_EARLY & 128)
  return make_uint4(X.lane * 0x01010101u + chunk, X.lane * 0x3010501u, chunk * 0x10101u, X.lane ^ chunk);   // TIMING EXPERIMENT ONLY: no HBM reads
#endif
  const uint32_t soff = live ? X.blk_off + (uint32_t)((X.vstart + 64 * chunk) * 16) : 0xfffff000u;
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(X.rsrc, X.lane * 16, soff, HRFD_STREAM_AUX);
  return make_uint4(v.x, v.y, v.z, v.w);
}

// phase difference -> de-emphasis numerator, WbFmDemodulator.cc:404-430 and the
// FIR half of IirFilter::filterData: d = wrap(theta - theta_prev); x = K*d;
// p = b0*x (b1 == b0, so p is also next sample's b1*x[n-1]); v = p + p_prev.
template <bool RINT = false>
__device__ __forceinline__ float numerator_p(float theta, float theta_prev, float kgain)
{
  float d = theta - theta_prev;
  d = RINT ? wrap_pi_rint(d) : wrap_pi(d);
  const float x = kgain * d;
  return DEEMPH_B0 * x;
}

// Run chunks [c0, c1) (64 samples each, chunk c covers positions vstart + 64c ..)
// on one wave and store v[pos] into LDS.  c0, c1, wlo, whi must be wave-uniform.
//
//  REPAIR == false (phase A proper): the run starts without knowing theta and
//    b0*x of the sample before it (another wave produces them concurrently), so
//    v of the run's first two samples is provisional; the four edge thetas
//    (first two, last two) are returned in edge[] and k_rx_wbfm patches the two
//    samples after the barrier.  A run that starts the stream (first block,
//    chunk 0) takes both from the carried state instead.
//  REPAIR == true: one extra, discarded, chunk in front re-creates the carries;
//    only positions [wlo, whi) are stored.
//
// Software pipeline with rotating registers: the main loop body is straight-line
// code for kDepth chunks (no branch, no copy of an in-flight load), so that the
// compiler can place counted s_waitcnt vmcnt(N).  The raw load of chunk i+kDepth
// is issued as soon as chunk i has been consumed, and the atan2 table gather of
// chunk i is in flight while chunk i-1 is finished.
// Timing ablations (tools/gpu_ab.py) are compile-time only, -DHRFD_ABLATE=<bit mask>: as
// run-time flags they put branches into the chunk loop and force vmcnt(0) waits.  Results
// are wrong when the mask is non-zero.
#ifndef HRFD_ABLATE
#define HRFD_ABLATE 0
#endif
__device__ __forceinline__ constexpr bool ablate(const RxParams &, int bit) { return (HRFD_ABLATE & bit) != 0; }

#ifndef HRFD_SCHED_FENCE
#define HRFD_SCHED_FENCE 1
#endif
#ifndef HRFD_DEPTH
#define HRFD_DEPTH 4
#endif
constexpr int kDepth = HRFD_DEPTH;     // raw chunks in flight per wave
#ifndef HRFD_LOOK
#define HRFD_LOOK 2
#endif
constexpr int kLook = HRFD_LOOK;    // chunks between issuing a table gather and using its result (< kDepth)

// The atan2 correction bytes and reciprocals (8.5 KiB, L2-resident) were requested at kernel
// entry, one 16-byte piece per thread; they go to LDS once the first raw chunks of the run
// are in flight, so that their latency and the raw stream's first miss overlap.  Every wave of
// the workgroup calls this exactly once (it contains the barrier).
__device__ __forceinline__ void publish_atan_tables(const StreamCtx &X)
{
  static_assert(kCorrBytes / 16 + kInvEntries / 4 <= kThreads, "one copy per thread");
  const int tid = threadIdx.x;
  if (tid < kCorrBytes / 16)
  {
    reinterpret_cast<uint4 *>(const_cast<uint8_t *>(X.atc))[tid] = X.tab;
  }
  else if (tid < kCorrBytes / 16 + kInvEntries / 4)
  {
    reinterpret_cast<uint4 *>(const_cast<float *>(X.ati))[tid - kCorrBytes / 16] = X.tab;
  }
  __syncthreads();
}

// calls f(std::integral_constant<int, r>) for the run-time residue r in [0, sizeof...(Rs))
template <typename F, int... Rs>
__device__ __forceinline__ void dispatch_residue(const int r, F &f, std::integer_sequence<int, Rs...>)
{
  ((r == Rs ? (f(std::integral_constant<int, Rs>{}), 0) : 0), ...);
}

template <int MODE, bool REPAIR, bool DUMP, bool S256, bool ARITH = false, int DEPTH = kDepth, int FENCE = HRFD_SCHED_FENCE>
__device__ __forceinline__ void produce_stream(const StreamCtx &X, const int c0, const int c1,
                                               const int wlo, const int whi, uint32_t &magsum,
                                               uint32_t (&edge)[4])
{
  // chunks between issuing a table gather and using its result; computed thetas need none
  constexpr int LOOK = ARITH ? 0 : kLook;
  const RxParams &P = *X.P;
  const int lane = X.lane;
  const MixConst mc = mix_const(lane & 3);               // position & 3 (chunks are 64-aligned)
  // REPAIR: one discarded chunk in front re-creates the carries -- except at the very start of
  // the stream (first block of the call, chunk 0), where the carried state does
  const int cbeg = REPAIR ? ((X.first && c0 == 0) ? 0 : c0 - 1) : c0;
  const bool lead = REPAIR && cbeg < c0;
  if (cbeg >= c1)
  {
    if (ARITH && !REPAIR && X.publish)
    {
      publish_atan_tables(X);
    }
    return;
  }
  FeCarry fc = {0x80808080u, 0x00800080u, 0x00800080u};   // zero input, zero stage outputs (offset binary)
  uint32_t c_theta = 0, c_p = 0;                         // lane 0: theta, b0*x of the sample before
  if (X.first && cbeg == 0)
  {
    // the stream continues from the previous call: carried state
    if (!S256)
    {
      fc = carry_from_16(*reinterpret_cast<const uint4 *>(X.st->fe_tail));
    }
    c_theta = f2u(X.st->wb_theta);
    c_p = f2u(X.st->wb_p);
  }
  else if (!S256)
  {
    // the three front-end carries depend on the 16 bytes before the run only
    const uint32_t soff = X.blk_off + (uint32_t)((X.vstart + 64 * cbeg) * 16 - 16);
    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(X.rsrc, 0, soff, 0);
    fc = carry_from_16(make_uint4(t.x, t.y, t.z, t.w));
  }
  uint16_t *dump = (DUMP && P.iq256 != nullptr)
                       ? reinterpret_cast<uint16_t *>(P.iq256 + X.ounit * (size_t)(2 * X.n256))
                       : nullptr;
  const float *lut = P.atan2_lut;
  const int nskip = (-X.vstart) >> 6;                    // chunks of history in front of the block: not in the squelch sum

  // stage 1: raw chunk -> 256 kS/s sample, side outputs, atan2 gather issued
  auto front = [&](const uint4 raw, const int ch) -> float {
    uint32_t y3 = 0, mixed;
    if (S256)
    {
      const uint32_t w = raw.x ^ 0x8080u;                // int8 -> (value + 128)
      mixed = (w & 0xffu) | ((w & 0xff00u) << 8);
    }
    else
    {
      if (ablate(P, 32))                                 // TIMING EXPERIMENT ONLY: no front-end arithmetic
      {
        y3 = raw.x ^ raw.y ^ raw.z ^ raw.w;
        mixed = y3 & 0x00ff00ffu;
      }
      else
      {
        y3 = frontend(raw, fc);
        mixed = mix_fs4(y3, mc);                         // (q_idx << 16) | i_idx
      }
    }
    if (!REPAIR && !S256)
    {
      const uint32_t mag = magnitude(mixed);             // rotation invariant: shares |i|, |q| with theta_arith
      magsum += (ch >= nskip) ? mag : 0u;
      if (DUMP && dump != nullptr && ch >= nskip)        // optional `enable iqdump` stream
      {
        const uint32_t iq = mixed ^ 0x00800080u;
        dump[X.vstart + 64 * ch + lane] = (uint16_t)((iq & 0xffu) | ((iq >> 8) & 0xff00u));
      }
    }
    if (MODE == 1 || MODE == 2 || MODE == 4)
    {
      // AM / FM / SSB: the mixed 256 kS/s sample as two int16 rails in LDS
      int16_t *rails = reinterpret_cast<int16_t *>(X.lds);
      const int at = X.vstart + 64 * ch + lane + X.hal;
      rails[at] = (int16_t)((int)(mixed & 0xffu) - 128);
      rails[X.qoff + at] = (int16_t)((int)(mixed >> 16) - 128);
    }
    if (MODE != 3)
    {
      return 0.0f;
    }
    if (ARITH)
    {
      return theta_arith(mixed, X.atc, X.ati);
    }
    uint32_t idx = __builtin_amdgcn_perm(0u, mixed, 0x0c0c0200u);   // (q_idx << 8) | i_idx
    if (ablate(P, 1))                                    // TIMING EXPERIMENT ONLY: coalesced fake index
    {
      idx = (idx & 0xff00u) | (uint32_t)lane;
    }
    if (ablate(P, 16))                                   // TIMING EXPERIMENT ONLY: no gather at all
    {
      return u2f(idx);
    }
    return lut[idx];
  };
  // stage 2: phase difference, +-pi wrap, gain, de-emphasis numerator, LDS store
  auto finish = [&](const float theta, const int ch, const bool store) {
    if (ablate(P, 8))                                    // TIMING EXPERIMENT ONLY: no theta-domain math
    {
      X.lds[X.vstart + 64 * ch + lane + X.hal] = f2u(theta);
      return;
    }
    const float thp = u2f(shr1(f2u(theta), c_theta));
    c_theta = ror1(f2u(theta));
    const float p = numerator_p(theta, thp, X.kgain);
    const float pp = u2f(shr1(f2u(p), c_p));
    c_p = ror1(f2u(p));
    const float v = p + pp;
    const int pos = X.vstart + 64 * ch + lane;
    if (REPAIR ? (store && pos >= wlo && pos < whi) : true)
    {
      X.lds[pos + X.hal] = f2u(v);
    }
  };

  const int n = c1 - cbeg;                               // chunks in this run, >= 1
  if (n <= LOOK)
  {
    // a run too short for the pipeline: one chunk at a time
    for (int j = 0; j < n; j++)
    {
      const float t = front(load_chunk<S256>(X, cbeg + j, c1), cbeg + j);
      if (MODE == 3)
      {
        finish(t, cbeg + j, j == 0 ? !lead : true);
        if (!REPAIR && j == 0)
        {
          edge[0] = (uint32_t)__builtin_amdgcn_readlane((int)f2u(t), 0);
          edge[1] = (uint32_t)__builtin_amdgcn_readlane((int)f2u(t), 1);
        }
        if (!REPAIR && j == n - 1)
        {
          edge[2] = (uint32_t)__builtin_amdgcn_readlane((int)f2u(t), 62);
          edge[3] = (uint32_t)__builtin_amdgcn_readlane((int)f2u(t), 63);
        }
      }
    }
    return;
  }
  // The pipelined path.  Everything up to the main loop is branch-free so that the
  // compiler's s_waitcnt bookkeeping reaches the loop header with exactly the pending
  // loads the loop body leaves behind (a conditional prologue makes it drain the
  // pipeline once per loop iteration).
  float th[DEPTH];
  uint4 q[DEPTH];
  // chunk j lives in slot (j - cbeg) % DEPTH, both for its raw data and its theta
#pragma unroll
  for (int k = 0; k < DEPTH; k++)
  {
    q[k] = load_chunk<S256>(X, cbeg + k, c1);
    th[k] = 0.0f;
  }
  if (ARITH && !REPAIR && X.publish)
  {
    publish_atan_tables(X);
  }
  // prologue: LOOK fronts, nothing finished yet
#pragma unroll
  for (int k = 0; k < LOOK; k++)
  {
    th[k] = front(q[k], cbeg + k);
    q[k] = load_chunk<S256>(X, cbeg + k + DEPTH, c1);
  }
  // first step, peeled: it yields the leading edge
  th[LOOK % DEPTH] = front(q[LOOK % DEPTH], cbeg + LOOK);
  q[LOOK % DEPTH] = load_chunk<S256>(X, cbeg + LOOK + DEPTH, c1);
  if (MODE == 3)
  {
    finish(th[0], cbeg, !lead);
    if (!REPAIR)
    {
      edge[0] = (uint32_t)__builtin_amdgcn_readlane((int)f2u(th[0]), 0);
      edge[1] = (uint32_t)__builtin_amdgcn_readlane((int)f2u(th[0]), 1);
    }
  }
  // main loop: front chunk fr+k (slot (LOOK+1+k) % DEPTH), finish chunk fr+k-LOOK
  // (slot (1+k) % DEPTH); branch-free groups of DEPTH
  int fr = cbeg + LOOK + 1;
  for (; fr + DEPTH <= c1; fr += DEPTH)
  {
#pragma unroll
    for (int k = 0; k < DEPTH; k++)
    {
      const int sf = (LOOK + 1 + k) % DEPTH;
      th[sf] = front(q[sf], fr + k);
      q[sf] = load_chunk<S256>(X, fr + k + DEPTH, c1);            // refill this slot
      if (MODE == 3)
      {
        finish(th[(1 + k) % DEPTH], fr + k - LOOK, true);
      }
      if (FENCE > 0 && (k + 1) % (FENCE > 0 ? FENCE : 1) == 0)
      // one chunk at a time: without the fence the scheduler hoists the front ends of all
      // DEPTH chunks to the top of the loop body, which needs every prefetched chunk at once
      // (s_waitcnt vmcnt(0)) and leaves the refills a few instructions of lead time
      {
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // remainder: 0..DEPTH-1 more fronts, each with its finish
#pragma unroll
  for (int k = 0; k < DEPTH - 1; k++)
  {
    if (fr + k < c1)
    {
      const int sf = (LOOK + 1 + k) % DEPTH;
      th[sf] = front(q[sf], fr + k);
      if (MODE == 3)
      {
        finish(th[(1 + k) % DEPTH], fr + k - LOOK, true);
      }
    }
  }
  // drain: the last min(LOOK, n-1) chunks are fronted but not finished.  Their slots depend
  // on n mod DEPTH; one specialisation per residue keeps every slot index a constant (a
  // run-time select chain makes the compiler spill th[] to scratch).
  auto drain = [&](auto residue) {
    constexpr int R = decltype(residue)::value;          // n % DEPTH
#pragma unroll
    for (int d = LOOK; d >= 1; d--)
    {
      if (n - d >= 1)
      {
        finish(th[(R - d + 2 * DEPTH) % DEPTH], c1 - d, true);
      }
    }
    if (!REPAIR)
    {
      const float th_last = th[(R - 1 + DEPTH) % DEPTH];
      edge[2] = (uint32_t)__builtin_amdgcn_readlane((int)f2u(th_last), 62);
      edge[3] = (uint32_t)__builtin_amdgcn_readlane((int)f2u(th_last), 63);
    }
  };
  if (MODE == 3)
  {
    const int res = n % DEPTH;
    dispatch_residue(res, drain, std::make_integer_sequence<int, DEPTH>{});
  }
}

// Derives the correction bytes of theta_arith() from the reference table itself.
// One thread per (a, b): for each of the four octant classes that exist for it,
// fix = table - approx (in ulps) must lie in -2..1 (a 2-bit two's complement field); the mirrored entry (q < 0) must be
// the exact negation.  bad[0] counts violations (then the gather kernel is used).
// TAB: `inv` is the first-octant table T0 and the approximation is atan2_from_t0 (theta_tab)
template <bool TAB>
__global__ void k_build_atan_corr(const float *lut, const float *inv, uint8_t *corr, uint32_t *bad)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= kCorrBytes)
  {
    return;
  }
  if (t >= kTriEntries)
  {
    corr[t] = 0;
    return;
  }
  // t = a (a + 1) / 2 + b
  int a = (int)((__builtin_sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while ((a + 1) * (a + 2) / 2 <= t) a++;
  while (a * (a + 1) / 2 > t) a--;
  const int b = t - a * (a + 1) / 2;
  uint32_t byte = 0, nbad = 0;
  for (int v = 0; v < 4; v++)
  {
    const bool swap = (v & 1) != 0, negi = (v & 2) != 0;
    const int mi = swap ? b : a, mq = swap ? a : b;      // |i|, |q|
    if (swap && !(b < a))
    {
      continue;                                          // ties are classified swap = 0
    }
    const int i = negi ? -mi : mi;
    if (negi ? (mi == 0) : (mi == 128))
    {
      continue;                                          // i = -0 / i = +128 do not exist
    }
    const AtanApprox ap = TAB ? atan2_from_t0(inv[t], swap, negi) : atan2_approx((uint32_t)a, (uint32_t)b, swap, negi, inv[a]);
    bool have = false;
    for (int sgn = 0; sgn < 2; sgn++)
    {
      const int q = sgn ? -mq : mq;
      if (q > 127 || (sgn && mq == 0))
      {
        continue;
      }
      uint32_t ex = f2u(lut[(q + 128) * 256 + (i + 128)]);
      if (sgn)
      {
        ex ^= 0x80000000u;                               // table must be odd in q
      }
      const int32_t fix = (int32_t)(ex - f2u(ap.theta0));  // exact - approx in ulps
      if (fix < -2 || fix > 1)
      {
        nbad++;
        continue;
      }
      const int32_t code = fix & 3;
      if (have && ((byte >> ap.shift) & 3u) != (uint32_t)code)
      {
        nbad++;
      }
      byte |= (uint32_t)code << ap.shift;
      have = true;
    }
  }
  corr[t] = (uint8_t)byte;
  if (nbad)
  {
    atomicAdd(bad, nbad);
  }
}

// test hook (hrfd_rx_debug_atan_eval): theta_arith over the whole (q, i) domain, table layout
template <bool TAB>
__global__ void k_atan_eval(const uint8_t *corr, const float *inv, float *out)
{
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;       // (q_idx << 8) | i_idx
  if (t < 65536u)
  {
    const uint32_t mixed = ((t >> 8) << 16) | (t & 0xffu);
    out[t] = TAB ? theta_tab(mixed, corr, inv) : theta_arith(mixed, corr, inv);
  }
}

// test hook (hrfd_rx_debug_atan_eval_quad): theta_quad over the whole (q, i) domain, table layout
__global__ void k_atan_eval_quad(const uint32_t *tq, float *out)
{
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;       // (q_idx << 8) | i_idx
  if (t < 65536u)
  {
    // the ring's word: bytes (i, q) of sample 0 in the low half, of sample 1 in the high half; signed = offset binary ^ 0x80
    const uint32_t w = (t | (t << 16)) ^ 0x80808080u;
    const uint32_t a = abs4_s8(w);
    const float t0 = theta_quad<0>(w, a, tq), t1 = theta_quad<1>(w, a, tq);
    out[t] = (f2u(t0) == f2u(t1)) ? t0 : __builtin_nanf("");
  }
}

// c^n for the de-emphasis pole c = -a1 (approximate: only seeds use it)
__device__ __forceinline__ float deemph_pow(int n)
{
  return exp2f((float)n * -0.07517338f);                 // log2(0.9492274)
}

// ---- phase B: the de-emphasis recurrence of one block, in place (v -> y) ------------------------
// Every wave of k_rx_wbfm's workgroup calls it (`sync` is the workgroup barrier); the waves
// 0..kBWaves-1 own the tiles.  See the comment in k_rx_wbfm for the scheme (partial sums, seeds,
// warm-up, tiles, verification, repair).  hoff = index of position 0 in lds[].
struct RecurShared
{
  float *parr;                    // [kMaxTiles + 8] partial sums; in the repair path the speculated starts
  float *wfin;                    // [kBWaves] final y of each wave's last lane
  unsigned long long *badmask;    // [kBWaves]
  uint32_t *anybad;               // zeroed by the caller before the first sync of the block
  const float *yanchor;           // true y in front of tile j0's warm-up (continuation blocks)
};

template <int MODE, bool S256, bool ARITH, bool PERBLOCK, class Sync>
__device__ __forceinline__ void recurrence_phase(const RxParams &P, const StreamCtx &X, uint32_t *lds, const int hoff,
                                                 const RecurShared &R, const bool first, const bool cont,
                                                 const int wave, const int lane, Sync &&sync)
{
  constexpr int T = kTile;
  const int wt = P.warm_tiles, M = P.seed_terms;
  const int W = wt * T;
  const int ntiles = P.ntiles, origin = P.origin;
  const int j0 = (-origin) / T;                          // the tile that contains (or begins at) position 0
  const int fa = (first || cont) ? j0 : (wt + M);        // first lane that runs; it is anchored
  const float a1 = DEEMPH_A1;
  const bool bwave = wave < kBWaves;
  const int ti = wave * 64 + lane;                       // this lane's tile (waves >= kBWaves have none)
  const int s = origin + ti * T;
  const ChanState *st = X.st;
  float y = 0.0f, y_spec = 0.0f;
  int kskip_tile = 0;
  // B1: partial sums  P_i = sum_k c^k v[s + T-1 - k], c = -a1, as two interleaved Horner chains in c^2
  // (tiles that lie in front of this buffer's history slot are never needed)
  if (bwave && M > 0 && ti < ntiles && s + hoff >= 0 && !ablate(P, 2))
  {
    const float cc = -a1;
    const float c2 = cc * cc;
    const uint2 *p2 = reinterpret_cast<const uint2 *>(lds + (s + hoff));
    float pa = 0.0f, pb = 0.0f;
#pragma unroll 7
    for (int j = 0; j < T / 2; j++)
    {
      const uint2 w = p2[j];
      pa = __builtin_fmaf(pa, c2, u2f(w.x));
      pb = __builtin_fmaf(pb, c2, u2f(w.y));
    }
    float p = __builtin_fmaf(pa, cc, pb);
    if (first && ti == j0)
    {
      p += deemph_pow(s + T) * st->wb_y;                 // the stream's past, as seen from the end of this tile
    }
    R.parr[ti] = p;
  }
  sync();                                                // partial sums visible; every read of the v tail is done
  const bool active = bwave && ti >= fa && ti < ntiles && !ablate(P, 2);   // (flag 2: TIMING EXPERIMENT ONLY, skip phase B)
  if (active)
  {
    // the recurrence is a long dependent chain that needs few issue slots: in k_rx_wbfm let it
    // win arbitration against the streaming waves of the neighbouring workgroup
    if (PERBLOCK)
    {
      __builtin_amdgcn_s_setprio(3);
    }
    if (M > 0)
    {
      // y at the end of tile ti - wt - 1:  sum_{m < M} (c^T)^m P[ti - wt - 1 - m], oldest first
      float acc = 0.0f;
      for (int m = M; m >= 1; m--)
      {
        acc = __builtin_fmaf(acc, P.seed_ct, R.parr[ti - wt - m]);
      }
      y = acc;
    }
    int kskip = 0;
    if (first)
    {
      if (s <= W)
      {
        // the start lies at or before the stream start: carried y, steps at n < 0 skipped
        y = st->wb_y;
        kskip = W - s;
      }
    }
    else if (cont && ti == j0)
    {
      y = *R.yanchor;                                    // true y[s - W - 1], saved by the previous block
    }
    const uint32_t *vp = lds + (s - W + hoff);           // lane stride T = 2 (mod 4): 64-bit accesses, no bank conflicts
    if (first && wave == 0)
    {
      y = iir_run<false, true>(vp, nullptr, W, kskip, y);
    }
    else
    {
      y = iir_run<false, false>(vp, nullptr, W, 0, y);
    }
    y_spec = y;
    kskip_tile = kskip - W;
    if (PERBLOCK)
    {
      __builtin_amdgcn_s_setprio(0);
    }
  }
  // The first lanes of a wave warm up over tiles of the wave before it: nobody may overwrite v
  // with y before every warm-up has read it (without this barrier a wave that runs 70 steps ahead
  // spoils its neighbour's speculated start -- caught by the check, but a repair each time).
  sync();
  if (active)
  {
    if (PERBLOCK)
    {
      __builtin_amdgcn_s_setprio(3);
    }
    uint32_t *yp = lds + (s + hoff);
    if (first && wave == 0)
    {
      y = iir_run<true, true>(yp, yp, T, kskip_tile, y);
    }
    else
    {
      y = iir_run<true, false>(yp, yp, T, 0, y);
    }
    if (PERBLOCK)
    {
      __builtin_amdgcn_s_setprio(0);
    }
    if (lane == 63)
    {
      R.wfin[wave] = y;
      if (PERBLOCK && P.dbg != nullptr)
      {
        P.dbg[(size_t)blockIdx.x * kDbgSlots + 24 + wave] = __builtin_readcyclecounter();
      }
    }
  }
  sync();                                                // all chains done
  // B3: every lane but the anchored one checks its speculated start against its left neighbour's end
  if (bwave)
  {
    float y_left = u2f(shr1(f2u(y), f2u(y)));
    if (lane == 0 && wave > 0)
    {
      y_left = R.wfin[wave - 1];
    }
    const bool bad = active && ti > fa && !same_trajectory(y_left, y_spec);
    const unsigned long long bm = __ballot(bad);
    if (lane == 0)
    {
      R.badmask[wave] = bm;
      if (bm != 0ull)
      {
        *R.anybad = 1u;
      }
    }
  }
  sync();
  if (*R.anybad != 0u)
  {
    // rare: repair in ascending order.  A tile's true start is the y in front of it in the
    // buffer (its left neighbour is final by then); the speculated starts go to parr[].
    if (bwave && ti < ntiles)
    {
      R.parr[ti] = y_spec;
    }
    sync();
    if (wave == 0)
    {
      unsigned long long bm[kBWaves];
#pragma unroll
      for (int w = 0; w < kBWaves; w++)
      {
        bm[w] = R.badmask[w];
      }
      uint32_t repairs = 0;
#pragma unroll
      for (int w = 0; w < kBWaves; w++)
      {
        while (bm[w] != 0ull)
        {
          const int l = __ffsll((long long)bm[w]) - 1;   // wave-uniform
          bm[w] &= ~(1ull << l);
          repairs++;
          const int j = w * 64 + l;
          const int sj = origin + j * T;
          // re-derive v over tile j (it was overwritten by the mis-started y)
          const int rc0 = (sj - X.vstart) >> 6;
          const int rc1 = (sj + T - X.vstart + 63) >> 6;
          uint32_t dummy_mag = 0, dummy_e[4];
          produce_stream<MODE, true, false, S256, ARITH>(X, rc0, rc1, sj, sj + T, dummy_mag, dummy_e);
          if (lane == 0)
          {
            uint32_t *rp = lds + (sj + hoff);
            const float y_true = u2f(rp[-1]);
            iir_run<true, false>(rp, rp, T, 0, y_true);
          }
          // the right neighbour's speculation must now match the corrected final y
          if (j + 1 < ntiles)
          {
            const float yj = u2f(lds[sj + T - 1 + hoff]);
            const float sp = R.parr[j + 1];
            if (!same_trajectory(yj, sp))
            {
              if (l == 63)
              {
                if (w + 1 < kBWaves)
                {
                  bm[w + 1 < kBWaves ? w + 1 : w] |= 1ull;
                }
              }
              else
              {
                bm[w] |= 1ull << (l + 1);
              }
            }
          }
        }
      }
      if (lane == 0)
      {
        atomicAdd(&P.counters[kCntRepair], repairs);
      atomicAdd(&P.sticky[kCntTotRepair], repairs);
      }
    }
    sync();
  }
}


template <int MODE, bool S256, bool ARITH>
__global__ __launch_bounds__(kThreads, 8) void k_rx_wbfm(const RxParams P)
{
  __shared__ __attribute__((aligned(16))) uint32_t lds[kMaxNV];
  // arithmetic atan2 tables (zero-sized in the gather build of the kernel)
  __shared__ __attribute__((aligned(16))) uint8_t atcorr[ARITH ? kCorrBytes : 16];
  __shared__ __attribute__((aligned(16))) float atinv[ARITH ? kInvEntries : 4];
  static_assert(sizeof(uint32_t) * kMaxNV + kCorrBytes + sizeof(float) * kInvEntries + 1936 <= 81920,
                "two workgroups per CU need <= 80 KiB of LDS each");
  __shared__ uint32_t red[kWaves];
  __shared__ int8_t dbfs8[128];         // the reachable part of the dBFS table (0..48), so that the
                                        // squelch decision after phase A does not wait for a global load
  __shared__ float tailcarry[2];        // theta, b0*x of the block's last sample
  __shared__ uint32_t edges[kWaves][4]; // per run: theta of its first two and last two samples
  // phase B
  __shared__ float parr[kMaxTiles + 8]; // per-tile geometric partial sums; in the repair path the speculated starts
  __shared__ float wfin[kBWaves];       // final y of each recurrence wave's last lane
  __shared__ unsigned long long badmask[kBWaves];
  __shared__ uint32_t anybad;
  __shared__ float yanchor;             // true y in front of the warm-up of tile j0 of the run's NEXT block
  __shared__ float chk_prev;            // this block's cross-block check value, for the next block of the run
  __shared__ __attribute__((aligned(4))) int16_t ctail[kWbS + kWbU + kWbV + 2];   // integer-stage histories, ditto

  // A workgroup owns a RUN of up to P.run_len consecutive blocks of one channel and walks them in
  // order.  Only the run's first block re-derives history in front of it (and is verified across
  // blocks by k_rx_epilogue); a continuation block carries everything exactly: the tail of v (one
  // register per thread, it has to survive phase B, which overwrites v with y in place), theta
  // and b0*x of the last sample, one true y for its first tile's start, the integer stages'
  // histories.
  uint32_t ci, run;
  if (!map_unit(blockIdx.x, P.n_list, P.n_runs, ci, run))
  {
    return;
  }
  const uint32_t c = P.chan_list[ci];
  const int n256 = (int)P.n256;
  const uint32_t b_first = run * P.run_len;
  const uint32_t b_end = min(P.n_blocks, b_first + P.run_len);
  if (P.run_len > 1 && ((blockIdx.x >> 8) & 1u) != 0u)
  {
    // the two workgroups that share a CU (ids w and w + 256) would otherwise walk their runs in
    // lockstep and sit in phase B at the same time
    for (int i = 0; i < P.stagger; i++)
    {
      __builtin_amdgcn_s_sleep(127);
    }
  }
  uint32_t keep0 = 0u;                                   // v[n256 - nkeep + tid]
  for (uint32_t b = b_first; b < b_end; b++)
  {
  // Everything derived from the thread index is re-derived per block behind an opaque copy:
  // otherwise the compiler hoists dozens of per-lane constants out of this loop, keeps them
  // alive across all phases and spills them.
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // wave-uniform: keep it scalar
  const bool first = (b == 0);
  const bool cont = (b > b_first);                       // the previous block of the run is ours
  const int hal = P.hal;
  // de-emphasis geometry (rx_launch): tile i = [origin + i*kTile, +kTile), the last one ends at n256
  constexpr int T = kTile;
  const int wt = P.warm_tiles, M = P.seed_terms;
  const int W = wt * T;
  const int nkeep = (wt + M + 1) * T;                    // v history a block with a true start needs (<= kKeepMax)
  if (cont)
  {
    __syncthreads();                                     // the previous block's phase C is done with the buffer
    if (tid < nkeep)
    {
      lds[hal - nkeep + tid] = keep0;
    }
  }
  else if (first && MODE == 3 && tid < nkeep)
  {
    lds[hal - nkeep + tid] = 0u;                         // nothing precedes the stream start: the seeds sum zeros
  }
  const ChanState *st = P.state + c;
  const ChanCfg cfg = P.cfg[c];
  const size_t unit = (size_t)c * P.n_blocks + b;                    // launch-local scratch index

  StreamCtx X;
  X.P = &P;
  // buffer descriptor over this channel's input of the whole call: raw (stride 0),
  // num_records in bytes, dword 3 = 0x00020000 (32-bit data format, gfx9 family)
  X.rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int8_t *>(P.iq + (uint64_t)c * P.ch_stride), 0,
      (int)(P.n_blocks * P.block_bytes), 0x00020000);
  X.blk_off = b * P.block_bytes;
  X.st = st;
  X.lds = lds;
  X.ounit = (size_t)c * P.out_blocks + P.out_b0 + b;                 // index in the caller's outputs
  // K = (gain/75000)*32767 in float, that order (WbFmDemodulator.cc:392-395)
  X.kgain = cfg.gain_wbfm / 75000.0f;
  X.kgain = X.kgain * 32767.0f;
  X.hal = hal;
  X.vstart = (first || cont || ablate(P, 256)) ? 0 : -hal;   // first position of v we produce (256: TIMING EXPERIMENT ONLY, no history)
  X.n256 = n256;
  X.lane = lane;
  X.first = first;
  X.atc = atcorr;
  X.ati = atinv;
  if (!cont && tid < 128)
  {
    dbfs8[tid] = (int8_t)P.dbfs[tid];                    // read after the barrier that ends phase A
  }
  X.tab = make_uint4(0u, 0u, 0u, 0u);
  X.publish = ARITH && !cont;                            // the atan2 tables go to LDS once per workgroup
  if (ARITH && !cont)
  {
    // request this thread's piece of the atan2 tables now; produce_stream publishes them to LDS
    if (tid < kCorrBytes / 16)
    {
      X.tab = reinterpret_cast<const uint4 *>(P.at_corr)[tid];
    }
    else if (tid < kCorrBytes / 16 + kInvEntries / 4)
    {
      X.tab = reinterpret_cast<const uint4 *>(P.at_inv)[tid - kCorrBytes / 16];
    }
  }
  const int8_t *blk = P.iq + (uint64_t)c * P.ch_stride + (uint64_t)b * P.block_bytes;

#define HRFD_STAMP(i)                                                         \
  if (P.dbg != nullptr && tid == 0)                                           \
  {                                                                           \
    P.dbg[(size_t)blockIdx.x * kDbgSlots + (i)] = __builtin_readcyclecounter();       \
  }
  HRFD_STAMP(0)
  if (P.dbg != nullptr && tid == 0)
  {
    // where did the dispatcher put this workgroup?  HW_REG_HW_ID (4), HW_REG_XCC_ID (20)
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    P.dbg[(size_t)blockIdx.x * kDbgSlots + 6] = ((unsigned long long)xcc << 32) | hw;
  }
  // ----------------------------------------------------------------- phase A
  // the block's chunks, dealt to the waves as contiguous, balanced runs
  const int nch = (n256 - X.vstart) >> 6;                // 1 KiB chunks to run
  const int cbase = nch / kWaves, cextra = nch % kWaves;
  const int c0 = wave * cbase + min(wave, cextra);
  const int c1 = c0 + cbase + (wave < cextra ? 1 : 0);
  uint32_t magsum = 0;
  {
    uint32_t e[4] = {0u, 0u, 0u, 0u};
    if (P.iq256 != nullptr)
    {
      produce_stream<MODE, false, true, S256, ARITH>(X, c0, c1, X.vstart, n256, magsum, e);
    }
    else
    {
      produce_stream<MODE, false, false, S256, ARITH>(X, c0, c1, X.vstart, n256, magsum, e);
    }
    if (MODE == 3 && lane < 4)
    {
      edges[wave][lane] = (lane == 0) ? e[0] : (lane == 1) ? e[1] : (lane == 2) ? e[2] : e[3];
    }
  }

  HRFD_STAMP(1)
  if (P.dbg != nullptr && lane == 0)
  {
    P.dbg[(size_t)blockIdx.x * kDbgSlots + 8 + wave] = __builtin_readcyclecounter();
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
  if (tid == 0)
  {
    anybad = 0u;                                         // read after phase B's verification barrier
  }
  __syncthreads();
  uint32_t total = 0;
  for (int w = 0; w < kWaves; w++)
  {
    total += red[w];
  }
  const uint32_t mean_mag = total / (uint32_t)n256;      // SignalDetector.cc:255
  // DbfsCalculator::convertMagnitudeToDbFs (:111-147) with a 7-bit full scale
  int32_t dbfs = (int32_t)dbfs8[min(mean_mag, 127u)] - 42;
  dbfs = (int32_t)((uint32_t)dbfs - P.gain_db);
  const bool present = dbfs >= cfg.threshold;
  // Squelch::run + SignalTracker::run: allowed = present || tracking.  For b > 0
  // the predecessor's `present` is not known here: the batch speculates "open"
  // and k_rx_epilogue verifies it.
  // (the inner demodulator API has no squelch: X::acceptIqData always demodulates)
  const bool allowed = S256 ? true : (first ? (present || st->tracking != 0) : true);
  if (tid == 0)
  {
    P.magnitude[X.ounit] = mean_mag;
    P.present[unit] = present ? 1 : 0;
  }

  const bool last = (b + 1 == P.n_blocks);
  ChanState *so = P.state_out + c;
  if (last && tid < 4 && !S256)
  {
    // front-end carry for the next call: the last 16 raw bytes of this block
    reinterpret_cast<uint32_t *>(so->fe_tail)[tid] =
        reinterpret_cast<const uint32_t *>(blk + P.block_bytes - 16)[tid];
  }
  if (MODE != 3)
  {
    continue;                                            // front end + squelch only: next block of the run
  }
  if (!allowed)
  {
    // demodulator untouched (state frozen).  Only a call's very first block can get here; in a
    // multi-block call that is a gate violation, the launch is not committed and the host replays
    // block by block, so the rest of this run is not worth producing.
    break;
  }

  // ----------------------------------------------------------------- phase B
  // y[n] = v[n] - a1*y[n-1] (IirFilter.cc:161-176 with one recursive tap), in place.
  // The recurrence rounds twice per step and cannot be re-associated, so it is run as up to 256
  // tiles of T samples in parallel, one per lane of the first kBWaves waves (one wave per SIMD).
  // Lane i first computes the geometric partial sum of v over its own tile; a lane's start
  // value at W = wt*T samples before its tile is the sum of M such partials -- y to a few ulp.
  // From there the float trajectory re-synchronises *bit for bit* with the true one during the
  // warm-up (contraction 0.949 per step).  That is verified, not assumed: every lane compares
  // its warmed-up y[s-1] with its left neighbour's final y; a tile that has not merged is
  // re-derived and re-run from the true value, then the check moves right (rare).
  //   first block of the call: lanes whose start precedes the stream start take the carried y
  //     and skip the steps before position 0 (exact).
  //   continuation block: v history and one true y (yanchor) are carried: lane j0 is exact.
  //   other blocks (b > 0, first of a run): history re-derived from the raw input; the first
  //     wt + M lanes cannot be seeded and do not run; lane wt + M is checked across blocks by
  //     k_rx_epilogue (y at -645 as computed here vs. by the predecessor).
  const int origin = P.origin;
  const int j0 = (-origin) / T;                          // the tile that contains (or begins at) position 0
  HRFD_STAMP(2)
  if (wave == 0)
  {
    // Patch the two provisional samples at the start of every run but the first
    // (produce_stream): they need theta of the two samples before the run, which
    // the neighbouring wave produced.  One lane per run boundary.  In a continuation block
    // the first run starts provisionally too: its two samples need the previous block's last
    // theta and b0*x, still in tailcarry[].
    const int nruns = min(nch, kWaves);
    if (cont && lane == 0)
    {
      const float tm1 = tailcarry[0], pm1 = tailcarry[1];
      const float t0 = u2f(edges[0][0]), t1 = u2f(edges[0][1]);
      const float p0 = numerator_p(t0, tm1, X.kgain);
      const float p1 = numerator_p(t1, t0, X.kgain);
      lds[0 + hal] = f2u(p0 + pm1);
      lds[1 + hal] = f2u(p1 + p0);
    }
    if (lane >= 1 && lane < nruns)
    {
      const int w = lane;
      const int sw = X.vstart + 64 * (w * cbase + min(w, cextra));   // first position of run w
      const float tm2 = u2f(edges[w - 1][2]), tm1 = u2f(edges[w - 1][3]);
      const float t0 = u2f(edges[w][0]), t1 = u2f(edges[w][1]);
      const float pm1 = numerator_p(tm1, tm2, X.kgain);
      const float p0 = numerator_p(t0, tm1, X.kgain);
      const float p1 = numerator_p(t1, t0, X.kgain);
      lds[sw + hal] = f2u(p0 + pm1);
      lds[sw + 1 + hal] = f2u(p1 + p0);
    }
    if (lane == 0)
    {
      const float tl2 = u2f(edges[nruns - 1][2]), tl1 = u2f(edges[nruns - 1][3]);
      tailcarry[0] = tl1;
      tailcarry[1] = numerator_p(tl1, tl2, X.kgain);
    }
  }
  __syncthreads();                                       // v is complete
  if (b + 1 < b_end)
  {
    // the next block of the run continues from this one: keep the tail of v (the chains below,
    // behind the next barrier, overwrite it with y)
    keep0 = (tid < nkeep) ? lds[n256 - nkeep + tid + hal] : 0u;
  }
  if (P.serial)
  {
    // exact replay path (n_blocks == 1): one lane, the whole block in order
    __syncthreads();
    if (tid == 0)
    {
      const float a1 = DEEMPH_A1;
      float ys = st->wb_y;
      for (int n = 0; n < n256; n++)
      {
        const float r = a1 * ys;
        ys = u2f(lds[n + hal]) - r;
        lds[n + hal] = f2u(ys);
      }
    }
    __syncthreads();
  }
  else
  {
    const RecurShared R = {parr, wfin, badmask, &anybad, &yanchor};
    recurrence_phase<MODE, S256, ARITH, true>(P, X, lds, hal, R, first, cont, wave, lane, [] { __syncthreads(); });
  }
  HRFD_STAMP(3)
  HRFD_STAMP(4)
  if (tid == 0)
  {
    // cross-block check values: position -645 precedes every history sample the
    // integer stages read (-644); a block that re-derived its history computed it in tile wt + M.
    const int chk = -kHist + 59;
    const float pub = u2f(lds[n256 + chk + hal]);
    P.chk_spec[unit] = first ? 0.0f : (cont ? chk_prev : u2f(lds[chk + hal]));
    P.chk_pub[unit] = pub;
    chk_prev = pub;
    if (b + 1 < b_end)
    {
      yanchor = u2f(lds[n256 + (origin + j0 * T) - W - 1 + hal]);   // the next block's y[s_j0 - W - 1]
    }
  }

  if (ablate(P, 4))                                     // TIMING EXPERIMENT ONLY: skip phase C
  {
    continue;
  }
  // ----------------------------------------------------------------- phase C
  // C1: s[n] = (int16_t)y[n] (WbFmDemodulator.cc:476), repacked in place as
  // int16 pairs at the bottom of the buffer: all reads, barrier, all writes.
  const bool carried = first || cont;                    // the integer stages' histories are carried, not re-derived
  const bool small_y = fabsf(X.kgain) * 3.3f < 2147483000.0f;
  const int smin = carried ? 0 : -kHist;
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
      // |y| <= |K| pi (unit DC gain, positive impulse response): below 2^31 the cast cannot overflow
      w = small_y ? pack_s16<false>(ya, yb) : pack_s16<true>(ya, yb);
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
  if (carried)
  {
    // histories of the three integer stages: from the carried state, or from the previous block of the run
    const int16_t *hs = first ? st->wb_s : ctail;
    const int16_t *hu = first ? st->wb_u : ctail + kWbS;
    const int16_t *hv = first ? st->wb_v : ctail + kWbS + kWbU;
    if (tid < kWbS / 2)
    {
      lds[((kHist - kWbS) >> 1) + tid] = reinterpret_cast<const uint32_t *>(hs)[tid];
    }
    if (tid < kWbU)
    {
      U16[kUHist - kWbU + tid] = (uint16_t)hu[tid];
    }
    if (tid < kWbV)
    {
      V16[kVHist - kWbV + tid] = (uint16_t)hv[tid];
    }
  }
  __syncthreads();

  // C2: U[m] = D(8,4)(S), WbFmDemodulator.cc:468-472; two outputs per thread
  {
    const int mmin = carried ? 0 : -kUHist;
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

  // C3: V[k] = D(12,4)(U);  C4: PCM[p] = D(40,2)(V)
  stage_d12(lds + kUOff, lds + kVOff, carried ? 0 : -kVHist, n256 >> 4, tid);
  __syncthreads();
  stage_d40(lds + kVOff, n256 >> 5, reinterpret_cast<uint32_t *>(P.pcm + X.ounit * (size_t)(n256 >> 5)), tid);

  HRFD_STAMP(5)
  // carried histories: for the next call, and for the next block of the run
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
  if (b + 1 < b_end)
  {
    if (tid < kWbS / 2)
    {
      reinterpret_cast<uint32_t *>(ctail)[tid] = lds[((n256 + kHist - kWbS) >> 1) + tid];
    }
    if (tid < kWbU)
    {
      ctail[kWbS + tid] = (int16_t)U16[kUHist + (n256 >> 2) - kWbU + tid];
    }
    if (tid < kWbV)
    {
      ctail[kWbS + kWbU + tid] = (int16_t)V16[kVHist + (n256 >> 4) - kWbV + tid];
    }
  }
  }  // blocks of the run
}

// =============================================================================
//  phase A, "quad" layout (k_rx_wbfm_flow): a lane owns FOUR consecutive groups
// =============================================================================
// produce_stream gives lane L of a wave the 16-byte group L of each 1 KiB chunk, so every value a
// stage needs from the previous group comes from the neighbouring lane: 10 DPP moves per chunk
// (slow-class instructions, tools/ubench/valu_rate.hip).  With 128 VGPRs per wave a lane can own
// the 64 contiguous bytes of FOUR consecutive groups of a 4 KiB piece and run the stages
// breadth-first over them: the previous group is in the lane's own registers for three groups
// out of four (10 DPP per 4 KiB instead of 40), the Fs/4 rotation of a group is a compile-time
// constant (position mod 4 == group), and the four v go to LDS in one 16-byte store.
// The loads are four dwordx4 per lane, 64 bytes apart across lanes (every 128-byte line is
// touched by all four of them back to back).  Same arithmetic, same carries, same edges as
// produce_stream; positions are counted in pieces of 256 samples.
template <int ROT>
__device__ __forceinline__ uint32_t mix_fs4_const(uint32_t y3)
{
  // rot 0 (I,Q), 1 (-Q,I), 2 (-I,-Q), 3 (Q,-I) in offset-binary index form, see mix_fs4.  Negation of an index is
  // (256 - idx) & 255; the fields of y3 may carry anything above their low byte (the mask comes last), so the packed
  // forms below act modulo 256 per field: one packed subtract negates both, one packed multiply-add (-1, +1) one of them.
  if (ROT == 0)
  {
    return y3 & 0x00ff00ffu;
  }
  if (ROT == 2)
  {
    return as_u32u(as_us2(0x01000100u) - as_us2(y3)) & 0x00ff00ffu;
  }
  const us2 sw = as_us2(__builtin_amdgcn_alignbit(y3, y3, 16));
  if (ROT == 1)
  {
    return as_u32u(sw * us2{(unsigned short)0xffff, (unsigned short)1} + us2{(unsigned short)256, (unsigned short)0}) & 0x00ff00ffu;
  }
  return as_u32u(sw * us2{(unsigned short)1, (unsigned short)0xffff} + us2{(unsigned short)0, (unsigned short)256}) & 0x00ff00ffu;
}

struct QuadCarry
{
  FeCarry fe;
  uint32_t theta, p;          // lane 0: theta and b0*x of the sample before the piece
};

struct NoHook
{
  __device__ __forceinline__ void operator()(const uint32_t (&)[4][4]) const {}
};
// The front half of a piece: raw[j] = the lane's group j (16 bytes) -> the four mixed 256 kS/s samples of the lane,
// (q_idx << 16) | i_idx in offset binary (three half-band stages per rail and the Fs/4 rotation: frontend() / mix_fs4()
// in the quad layout).  `raw_free` is called once the raw registers are dead (behind stage 1, whose outputs it gets
// so that it can make whatever it does depend on them): the caller may refill the registers there.
template <class Hook>
__device__ __forceinline__ void quad_front(const uint4 (&raw)[4], FeCarry &fe, uint32_t (&mixed)[4], Hook &&raw_free)
{
  uint32_t r[4][4];
#pragma unroll
  for (int j = 0; j < 4; j++)
  {
    r[j][0] = raw[j].x ^ 0x80808080u;
    r[j][1] = raw[j].y ^ 0x80808080u;
    r[j][2] = raw[j].z ^ 0x80808080u;
    r[j][3] = raw[j].w ^ 0x80808080u;
  }
  // stage 1 (bytes) and its outputs as 16-bit (I,Q) fields
  uint32_t y1[4][4];
  const uint32_t rm1_0 = shr1(r[3][3], fe.x7);
  fe.x7 = ror1(r[3][3]);
#pragma unroll
  for (int j = 0; j < 4; j++)
  {
    const uint32_t rm1 = (j == 0) ? rm1_0 : r[j > 0 ? j - 1 : 0][3];
    const uint32_t a01 = __builtin_amdgcn_perm(r[j][0], rm1, 0x07060302u);
    const uint32_t b01 = __builtin_amdgcn_perm(r[j][1], r[j][0], 0x05040100u);
    const uint32_t c01 = __builtin_amdgcn_perm(r[j][1], r[j][0], 0x07060302u);
    const uint32_t a23 = __builtin_amdgcn_perm(r[j][2], r[j][1], 0x07060302u);
    const uint32_t b23 = __builtin_amdgcn_perm(r[j][3], r[j][2], 0x05040100u);
    const uint32_t c23 = __builtin_amdgcn_perm(r[j][3], r[j][2], 0x07060302u);
    const uint32_t y01 = hb1_bytes(a01, b01, c01);
    const uint32_t y23 = hb1_bytes(a23, b23, c23);
    y1[j][0] = __builtin_amdgcn_perm(0u, y01, 0x0c010c00u);
    y1[j][1] = __builtin_amdgcn_perm(0u, y01, 0x0c030c02u);
    y1[j][2] = __builtin_amdgcn_perm(0u, y23, 0x0c010c00u);
    y1[j][3] = __builtin_amdgcn_perm(0u, y23, 0x0c030c02u);
  }
  raw_free(y1);
  // stage 2
  uint32_t y20b[4], y21[4];
  const uint32_t y1m1_0 = shr1(y1[3][3], fe.y13);
  fe.y13 = ror1(y1[3][3]);
#pragma unroll
  for (int j = 0; j < 4; j++)
  {
    const uint32_t y1m1 = (j == 0) ? y1m1_0 : y1[j > 0 ? j - 1 : 0][3];
    y20b[j] = form_ac(hb2_sum(y1m1, y1[j][0] << 1, y1[j][1]));
    y21[j] = form_ac(hb2_sum(y1[j][1], y1[j][2] << 1, y1[j][3]));
  }
  // stage 3, mixer
  const uint32_t y2m1_0 = shr1(y21[3], fe.y21);
  fe.y21 = ror1(y21[3]);
  const uint32_t y30 = hb3_sum(y2m1_0, y20b[0], y21[0]) >> 2;
  const uint32_t y31 = hb3_sum(y21[0], y20b[1], y21[1]) >> 2;
  const uint32_t y32 = hb3_sum(y21[1], y20b[2], y21[2]) >> 2;
  const uint32_t y33 = hb3_sum(y21[2], y20b[3], y21[3]) >> 2;
  mixed[0] = mix_fs4_const<0>(y30);
  mixed[1] = mix_fs4_const<1>(y31);
  mixed[2] = mix_fs4_const<2>(y32);
  mixed[3] = mix_fs4_const<3>(y33);
}

// one 4 KiB piece: raw[j] = the lane's group j (16 bytes) -> v[4]; returns the four thetas
// ARITH: 0 table gather from global memory, 1 theta_arith, 2 theta_tab (X.ati is then T0)
// iqb (optional): the lane's four mixed samples as the eight bytes i0 q0 i1 q1 i2 q2 i3 q3 of the 256 kS/s stream
template <int ARITH, class Hook = NoHook>
__device__ __forceinline__ void quad_piece(const uint4 (&raw)[4], QuadCarry &c, const StreamCtx &X,
                                           uint32_t (&vout)[4], float (&theta)[4], uint32_t &mag4, Hook &&raw_free = NoHook(),
                                           uint32_t *iqb = nullptr)
{
  uint32_t mixed[4];
  quad_front(raw, c.fe, mixed, raw_free);
  if (iqb != nullptr)
  {
    // (q_idx << 16 | i_idx) in offset binary -> signed bytes i, q of two samples per dword
    iqb[0] = __builtin_amdgcn_perm(mixed[1], mixed[0], 0x06040200u) ^ 0x80808080u;
    iqb[1] = __builtin_amdgcn_perm(mixed[3], mixed[2], 0x06040200u) ^ 0x80808080u;
  }
  mag4 = 0;
#pragma unroll
  for (int j = 0; j < 4; j++)
  {
    mag4 += magnitude(mixed[j]);
    if (ARITH == 2)
    {
      theta[j] = theta_tab(mixed[j], X.atc, X.ati);
    }
    else if (ARITH == 1)
    {
      theta[j] = theta_arith(mixed[j], X.atc, X.ati);
    }
    else
    {
      theta[j] = X.P->atan2_lut[__builtin_amdgcn_perm(0u, mixed[j], 0x0c0c0200u)];
    }
  }
  // phase difference -> de-emphasis numerator
  const float thp0 = u2f(shr1(f2u(theta[3]), c.theta));
  c.theta = ror1(f2u(theta[3]));
  float p[4];
  p[0] = numerator_p<ARITH == 2>(theta[0], thp0, X.kgain);
  p[1] = numerator_p<ARITH == 2>(theta[1], theta[0], X.kgain);
  p[2] = numerator_p<ARITH == 2>(theta[2], theta[1], X.kgain);
  p[3] = numerator_p<ARITH == 2>(theta[3], theta[2], X.kgain);
  const float pp0 = u2f(shr1(f2u(p[3]), c.p));
  c.p = ror1(f2u(p[3]));
  vout[0] = f2u(p[0] + pp0);
  vout[1] = f2u(p[1] + p[0]);
  vout[2] = f2u(p[2] + p[1]);
  vout[3] = f2u(p[3] + p[2]);
}


// =============================================================================
//  finish: per channel -- squelch tracker over the batch, verification of both speculations,
//  n_pcm / allowed outputs, and the commit of the channel's pending state when the channel
//  verified clean.  One wave per channel (one lane per block for the checks, the lanes copy the
//  mode's state section in parallel).  The verdict is the CHANNEL's: a closed gate or a failed
//  speculation in one channel does not keep the others from committing (chan_fail / chan_poison);
//  the host replays the failed channels only.
//  Called from k_rx_finish (one workgroup per channel) and from the tail of k_rx_wbfm_flow (the
//  last workgroup of a channel finishes it).
// =============================================================================
// Shaped for latency: one wave runs this while nothing else of the channel does (in k_rx_wbfm_flow the rest of the CU idles
// behind it), so every input is requested before the first is used -- one memory round trip for everything that
// depends on the channel number alone (two when the mode has to be read first), the pending state included, read
// before the verdict is known -- and the stores come last (finish_gather, finish_apply).  k_rx_wbfm_flow hands the
// inputs over from its own LDS when the channel was one workgroup's: no round trip at all (a load takes microseconds
// while the other CUs still saturate the memory system).  MODE >= 0: the caller knows the channel's mode.
template <int MODE>
struct FinishIn
{
  static constexpr int kSecDw = (MODE == 3) ? 1 : 6;
  int mode;
  uint32_t tracking, poison, expired;
  uint32_t pl_raw, pp_raw;      // `present` of the last block and of the one before
  uint32_t pres0;               // lane b: `present` of block b (the first 64 blocks)
  float spec0, pub0;            // lane b: the check values around the start of block b
  uint32_t fe;                  // lanes 0..3: the pending fe_tail
  uint32_t sec[kSecDw];         // the mode's section of the pending state, dword lane + 64 k
};

// byte offset and dwords of a mode's section of ChanState (contiguous, 4-byte aligned)
__device__ __forceinline__ void state_section(const int mode, int &off, int &nd)
{
  static_assert(offsetof(ChanState, wb_theta) % 4 == 0 && offsetof(ChanState, fm_tail) % 4 == 0 && offsetof(ChanState, am_tail) % 4 == 0 &&
                offsetof(ChanState, ssb_tail) % 4 == 0 && sizeof(ChanState) % 4 == 0, "state sections are copied as dwords");
  static_assert((offsetof(ChanState, fm_tail) - offsetof(ChanState, wb_theta)) / 4 <= 64 &&
                (offsetof(ChanState, am_tail) - offsetof(ChanState, fm_tail)) / 4 <= 6 * 64 &&
                (offsetof(ChanState, ssb_tail) - offsetof(ChanState, am_tail)) / 4 <= 6 * 64 &&
                (sizeof(ChanState) - offsetof(ChanState, ssb_tail)) / 4 <= 6 * 64, "section sizes");
  off = 0;
  nd = 0;
  if (mode == 3) { off = (int)offsetof(ChanState, wb_theta); nd = ((int)offsetof(ChanState, fm_tail) - off) / 4; }
  else if (mode == 2) { off = (int)offsetof(ChanState, fm_tail); nd = ((int)offsetof(ChanState, am_tail) - off) / 4; }
  else if (mode == 1) { off = (int)offsetof(ChanState, am_tail); nd = ((int)offsetof(ChanState, ssb_tail) - off) / 4; }
  else if (mode == 4 || mode == 5) { off = (int)offsetof(ChanState, ssb_tail); nd = ((int)sizeof(ChanState) - off) / 4; }
}

template <int MODE>
__device__ __forceinline__ void finish_gather(const EpilogueParams &E, const uint32_t c, const int lane, FinishIn<MODE> &I)
{
  const ChanState *dst = E.state + c;
  const ChanState *src = E.state_out + c;
  const uint32_t nb = E.n_blocks;
  const uint8_t *pres = E.present + (size_t)c * nb;
  // ---- round trip 1
  I.mode = MODE;
  if (MODE < 0)
  {
    I.mode = E.cfg[c].mode;
  }
  I.tracking = dst->tracking;
  I.poison = E.chan_poison[c];
  I.expired = E.chan_expired[c];
  I.pl_raw = pres[nb - 1];
  I.pp_raw = pres[nb >= 2 ? nb - 2 : 0];
  const bool in0 = (uint32_t)lane < nb;
  I.pres0 = in0 ? (uint32_t)pres[lane] : 0u;
  I.spec0 = 0.0f;
  I.pub0 = 0.0f;
  if (in0 && lane > 0 && (MODE < 0 || MODE == 3))
  {
    I.spec0 = E.chk_spec[(size_t)c * nb + lane];
    I.pub0 = E.chk_pub[(size_t)c * nb + lane - 1];
  }
  I.fe = (lane < 4) ? reinterpret_cast<const uint32_t *>(src->fe_tail)[lane] : 0u;
  // ---- round trip 2 (the same one when the mode is known): the mode's section of the pending state
  int off, nd;
  state_section(I.mode, off, nd);
  const uint32_t *ssec = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(src) + off);
#pragma unroll
  for (int k = 0; k < FinishIn<MODE>::kSecDw; k++)
  {
    I.sec[k] = (lane + 64 * k < nd) ? ssec[lane + 64 * k] : 0u;
  }
}

template <int MODE>
__device__ __forceinline__ void finish_apply(const EpilogueParams &E, const uint32_t c, const int lane, const FinishIn<MODE> &I)
{
  ChanState *dst = E.state + c;
  const uint32_t nb = E.n_blocks;
  const uint8_t *pres = E.present + (size_t)c * nb;
  const int mode = I.mode;
  const uint32_t tracking = I.tracking, poison = I.poison, expired = I.expired, pl_raw = I.pl_raw, pp_raw = I.pp_raw;
  const uint32_t pres0 = I.pres0, fe = I.fe;
  const float spec0 = I.spec0, pub0 = I.pub0;
  constexpr int kSecDw = FinishIn<MODE>::kSecDw;
  int off, nd;
  state_section(mode, off, nd);
  uint32_t *dsec = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(dst) + off);

  // ---- the verdict: squelch tracker over the batch (Squelch.cc:227-273, SignalTracker.cc:104-146), both speculations
  uint32_t carry = tracking != 0 ? 1u : 0u;               // `present` of the block before
  uint32_t gate_viol = 0, spec_viol = 0;
  for (uint32_t b0 = 0; b0 < nb; b0 += 64)
  {
    const uint32_t b = b0 + lane;
    const bool in = b < nb;
    const size_t unit = (size_t)c * nb + b;
    const size_t ounit = (size_t)c * E.out_blocks + E.out_b0 + b;
    uint32_t present = pres0 != 0u ? 1u : 0u;
    float spec = spec0, pub = pub0;
    if (b0 != 0)                                          // more than 64 blocks: rare, not shaped
    {
      present = in ? (uint32_t)(pres[b] != 0) : 0u;
      if (in && mode == 3)
      {
        spec = E.chk_spec[unit];
        pub = E.chk_pub[unit - 1];
      }
    }
    const uint32_t prev = shr1(present, carry);           // lane 0 <- carried
    const bool allowed = (present | prev) != 0;
    if (in)
    {
      const bool demod = allowed && mode != 0;
      if (E.allowed != nullptr)
      {
        E.allowed[ounit] = allowed ? 1 : 0;
      }
      if (E.n_pcm != nullptr)
      {
        E.n_pcm[ounit] = demod ? E.n_pcm_per_block : 0u;
      }
      // the batch assumed every gate open
      gate_viol += (nb > 1 && mode != 0 && !allowed) ? 1u : 0u;
      if (b > 0 && mode == 3)
      {
        spec_viol += same_trajectory(spec, pub) ? 0u : 1u;
      }
    }
    carry = (uint32_t)__builtin_amdgcn_readlane((int)present, 63);
  }
  for (int o = 32; o > 0; o >>= 1)
  {
    gate_viol += __shfl_down(gate_viol, o);
    spec_viol += __shfl_down(spec_viol, o);
  }
  gate_viol = (uint32_t)__builtin_amdgcn_readfirstlane((int)gate_viol);
  spec_viol = (uint32_t)__builtin_amdgcn_readfirstlane((int)spec_viol);
  // a launch behind an unrepaired failed one started this channel from a stale state: it must not commit
  const uint32_t bits = (gate_viol ? kFailGate : 0u) | (spec_viol ? kFailSpec : 0u) |
                        (poison != 0u ? kFailPoison : 0u) | (expired != 0u ? kFailExpired : 0u);
  const bool clean = bits == 0u;
  if (lane == 0)
  {
    E.chan_fail[c] = bits;
    if (expired != 0u)
    {
      E.chan_expired[c] = 0u;
    }
    if (!clean)
    {
      E.chan_poison[c] = 1u;
      atomicAdd(&E.counters[kCntFail], 1u);
      atomicAdd(&E.sticky[kCntTotViol], 1u);
      if (gate_viol)
      {
        atomicAdd(&E.counters[kCntGate], gate_viol);
      }
      if (spec_viol)
      {
        atomicAdd(&E.counters[kCntSpec], spec_viol);
      }
    }
    if (c == E.first_channel)
    {
      atomicAdd(&E.sticky[kCntTotLaunch], 1u);
    }
  }
  if (c == E.first_channel && lane < kCntSticky)
  {
    E.next_local[lane] = 0u;                             // the next launch counts into the other set
  }
  if (!clean)
  {
    return;
  }
  // ---- commit: was the last block demodulated?  (else: demodulator state frozen)
  const bool p_last = pl_raw != 0u;
  const bool p_prev = (nb >= 2) ? (pp_raw != 0u) : (tracking != 0u);
  if (lane < 4)
  {
    reinterpret_cast<uint32_t *>(dst->fe_tail)[lane] = fe;
  }
  if (p_last || p_prev)
  {
#pragma unroll
    for (int k = 0; k < kSecDw; k++)
    {
      if (lane + 64 * k < nd)
      {
        dsec[lane + 64 * k] = I.sec[k];
      }
    }
  }
  if (lane == 0)
  {
    dst->tracking = p_last ? 1u : 0u;
  }
}

template <int MODE = -1>
__device__ __forceinline__ void finish_channel(const EpilogueParams &E, const uint32_t c, const int lane)
{
  FinishIn<MODE> I;
  finish_gather<MODE>(E, c, lane, I);
  finish_apply<MODE>(E, c, lane, I);
}

// the channels that no kernel finishes by itself: one wave per channel of E.chan_list (or 0 .. n_channels - 1)
__global__ __launch_bounds__(64) void k_rx_finish(const EpilogueParams E)
{
  if (blockIdx.x >= E.n_channels)
  {
    return;
  }
  const uint32_t c = (E.chan_list != nullptr) ? E.chan_list[blockIdx.x] : blockIdx.x;
  finish_channel<>(E, c, (int)threadIdx.x);
}

// explicit instantiations used by the host side
template __global__ void k_rx_wbfm<0, false, false>(const RxParams);
template __global__ void k_rx_wbfm<3, false, false>(const RxParams);
template __global__ void k_rx_wbfm<3, false, true>(const RxParams);
template __global__ void k_rx_wbfm<3, true, false>(const RxParams);

} // namespace hrfd
