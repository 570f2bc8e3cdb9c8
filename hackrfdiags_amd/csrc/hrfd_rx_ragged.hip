// hackrfdiags_amd/csrc/hrfd_rx_ragged.hip -- the receive chain for ANY block length (gfx950).
//
// IqDataProcessor::acceptIqData(ts, buffer, byteCount) takes whatever DataConsumer hands it: a short USB transfer is
// counted and passed on (DataConsumer.cc:229-241, :341-343), and every decimator of the chain keeps its commutator
// position between calls (Decimator_int16.cc:321-362), so the reference demodulates blocks of any length -- a call that
// ends between two outputs of a stage leaves the stage part way through a group of M inputs.  The streaming kernels
// (hrfd_rx_flow.hip, hrfd_rx_kernels.hip, hrfd_rx_fir_kernels.hip) are built on whole groups: they take blocks of a
// multiple of 1024 bytes on a stream whose blocks all were multiples of 512.  Everything else comes here:
//
//   k_rx_ragged   one workgroup per channel, the blocks of the call one after the other, every stage as the plain
//                 D(N, M, h) of SURVEY 8a with a commutator position -- all outputs of a stage in parallel, the float
//                 recurrences on one lane, in the reference's order of operations.  Exact by construction: nothing is
//                 speculated, nothing verified, a channel always commits.
//   on the grid   (block a multiple of 512 bytes, handle never saw another length): the state is ChanState, read and
//                 written in the streaming kernels' format (decimator pipelines re-created from the demodulator's input
//                 tail the way k_rx_fir does) -- a 261632-byte block (a transfer one USB packet short) passes through
//                 here and the next 262144-byte block is back on the streaming kernels.
//   off the grid  (any other even length, once): the state is RagState (hrfd_device.h) from then on, built from
//                 ChanState by k_rag_expand; every later call of the handle runs here.
//
// Per call and channel the reference produces (verified against the compiled reference, tests/golden/make_golden_short.py):
//   decimatedByteCount = 2 * floor((p + byteCount / 2) / 8), p = IQ samples held by the front end's commutators
//   (IqDataProcessor.cc:429-500); the Fs/4 rotation restarts at every call (:771-815: index within the call); the
//   squelch mean divides by the call's own sample count (SignalDetector.cc:255); the demodulator is handed that many
//   bytes and emits floor((phase + n) / M) samples per stage.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hrfd_device.h"
#include "hrfd_tables.h"

namespace hrfd {

constexpr int kRagThreads = 1024;

// One Q15 stage D(N, M, h) over cnt new inputs: Decimator_int16::decimate / filterData (Decimator_int16.cc:176-249,
// :321-362), FirFilter_int16::filterData for M = 1 (FirFilter_int16.cc:151-224).  x points at the first new input;
// x[-(N-1) * STRIDE] .. x[-STRIDE] hold the stage's pipeline (its last N - 1 inputs).  `phase` inputs of the current
// group were shifted in by earlier calls: output k completes at input (M - 1 - phase) + k M.  All outputs in parallel;
// int32 wrap-around accumulation, rounding constant 1 << 14, arithmetic shift, low 16 bits.  Returns the output count.
template <int N, int M, typename TIn, int STRIDE>
__device__ __forceinline__ int rag_stage(const int16_t (&taps)[N], const int phase, const TIn *x, const int cnt,
                                         int16_t *y, const int tid)
{
  const int first = M - 1 - phase;
  const int nout = (cnt > first) ? (cnt - first + M - 1) / M : 0;
  for (int k = tid; k < nout; k += kRagThreads)
  {
    const TIn *p = x + (first + k * M) * STRIDE;
    uint32_t acc = 1u << 14;
#pragma unroll
    for (int t = 0; t < N; t++)
    {
      acc += (uint32_t)((int)taps[t] * (int)p[-t * STRIDE]);
    }
    y[k] = (int16_t)((int32_t)acc >> 15);
  }
  return nout;
}

// x[-L .. -1] <- x[cnt - L .. cnt - 1]: the pipeline for the next call (elements of T: both rails of an interleaved
// buffer at once).  The barrier inside also completes the stage that has just written its outputs.
template <typename T>
__device__ __forceinline__ void rag_keep(T *x, const int cnt, const int L, const int tid)
{
  T v = 0;
  if (tid < L)
  {
    v = x[cnt - L + tid];
  }
  __syncthreads();
  if (tid < L)
  {
    x[-L + tid] = v;
  }
}

struct RagCtx
{
  float *F;            // WBFM: theta -> v -> y of the block.  FM: F[0 .. 3] theta[n-4 .. n-1], F[4 + k] theta of tuner output k
  int8_t *mix;         // the call's 256 kS/s stream behind the Fs/4 mix, (i, q) bytes; in front of it the first stage's pipeline
  int16_t *s0;         // WBFM: (int16_t)y -- the SAME memory as mix (the bytes are dead by then); s0[-7 ..] D(8,4)'s pipeline
  int16_t *r1[2];      // input of the second stage (WBFM / FM: D(12,4), rail 0 only; AM / SSB: D(12,4) per rail)
  int16_t *r2[2];      // input of the third stage (D(40,2); D(16,2) per rail)
  int16_t *r3[2];      // SSB: 8 kS/s I (the negating delay's input) and Q (the Hilbert transformer's)
  int16_t *r4[2];      // SSB: their outputs
  uint8_t *hist8;      // on the grid: the active FIR demodulator's input tail (ChanState::fm_tail / am_tail / ssb_tail)
  int16_t *h8k[2];     // on the grid, SSB: ChanState::ssb_i / ssb_q
  float *fst;          // WBFM: theta, b1 x, y of the last sample; AM / SSB: dc-removal x[n-1], y[n-1]
  const float *lut;
  int ph[3];           // commutator positions of the active demodulator's three stages (uniform)
  int tid;
};

__device__ __forceinline__ int rag_kind(const int mode)
{
  return (mode == 1) ? 1 : (mode == 2) ? 2 : (mode == 3) ? 3 : (mode == 4 || mode == 5) ? 4 : 0;
}

__device__ __forceinline__ float rag_theta(const RagCtx &X, const int i, const int q)
{
  // atan2LookupTable[(uint8_t)q + 128][(uint8_t)i + 128] (WbFmDemodulator.cc:404-409; FmDemodulator.cc:495-499 on the low bytes)
  const uint32_t ii = ((uint32_t)i & 0xffu) ^ 0x80u;
  const uint32_t qi = ((uint32_t)q & 0xffu) ^ 0x80u;
  return X.lut[(qi << 8) | ii];
}

__device__ __forceinline__ void rag_zero_heads(RagCtx &X)
{
  const int tid = X.tid;
  if (tid < 2 * kRagHead)
  {
    X.mix[-2 * kRagHead + tid] = 0;
  }
  if (tid < kRagHead)
  {
    for (int r = 0; r < 2; r++)
    {
      X.r1[r][-kRagHead + tid] = 0;
      X.r2[r][-kRagHead + tid] = 0;
      X.r3[r][-kRagHead + tid] = 0;
    }
  }
  if (tid < 4)
  {
    X.F[tid] = 0.0f;
    X.fst[tid] = 0.0f;
  }
  __syncthreads();
}

// ---- ChanState -> the stages' pipelines (every commutator at 0).  The FIR demodulators keep their INPUT tail in
//      ChanState and re-create the pipelines from it (as k_rx_fir does at the start of a call): the tail is run
//      through the stages from zero pipelines; what is left in them depends on its last 684 / 324 samples only.
__device__ __forceinline__ void rag_load_chan(RagCtx &X, const ChanState *st, const int kind)
{
  const int tid = X.tid;
  X.ph[0] = X.ph[1] = X.ph[2] = 0;
  if (kind == 3)
  {
    if (tid == 0)
    {
      X.fst[0] = st->wb_theta;
      X.fst[1] = st->wb_p;
      X.fst[2] = st->wb_y;
    }
    if (tid < kWbS) X.s0[-kWbS + tid] = st->wb_s[tid];
    if (tid < kWbU) X.r1[0][-kWbU + tid] = st->wb_u[tid];
    if (tid < kWbV) X.r2[0][-kWbV + tid] = st->wb_v[tid];
  }
  else if (kind == 2)
  {
    for (int k = tid; k < 2 * kFmTail; k += kRagThreads)
    {
      const uint8_t b = st->fm_tail[k];
      X.hist8[k] = b;
      X.mix[k] = (int8_t)(b ^ 0x80u);
    }
    __syncthreads();
    const int n1 = rag_stage<N_FM_TUNER_D32, 4, int8_t, 2>(Q_FM_TUNER_D32, 0, X.mix, kFmTail, X.r1[0], tid);
    (void)rag_stage<N_FM_TUNER_D32, 4, int8_t, 2>(Q_FM_TUNER_D32, 0, X.mix + 1, kFmTail, X.r1[1], tid);
    rag_keep(X.mix, 2 * kFmTail, 2 * (N_FM_TUNER_D32 - 1), tid);
    if (tid < 4)
    {
      const int k = n1 - 4 + tid;
      X.F[tid] = rag_theta(X, X.r1[0][k], X.r1[1][k]);
    }
    __syncthreads();
    if (tid < kWbU) X.r1[0][-kWbU + tid] = st->fm_u[tid];
    if (tid < kWbV) X.r2[0][-kWbV + tid] = st->fm_v[tid];
  }
  else if (kind == 1 || kind == 4)
  {
    const uint8_t *tail = (kind == 1) ? st->am_tail : st->ssb_tail;
    for (int k = tid; k < 2 * kAmTail; k += kRagThreads)
    {
      const uint8_t b = tail[k];
      X.hist8[k] = b;
      X.mix[k] = (int8_t)(b ^ 0x80u);
    }
    __syncthreads();
    int n1 = 0, n2 = 0;
    for (int r = 0; r < 2; r++)
    {
      n1 = rag_stage<N_AM_D1, 4, int8_t, 2>(Q_AM_D1, 0, X.mix + r, kAmTail, X.r1[r], tid);
    }
    rag_keep(X.mix, 2 * kAmTail, 2 * (N_AM_D1 - 1), tid);
    for (int r = 0; r < 2; r++)
    {
      n2 = rag_stage<N_AM_D2, 4, int16_t, 1>(Q_AM_D2, 0, X.r1[r], n1, X.r2[r], tid);
    }
    rag_keep(X.r1[0], n1, N_AM_D2 - 1, tid);
    rag_keep(X.r1[1], n1, N_AM_D2 - 1, tid);
    for (int r = 0; r < 2; r++)
    {
      (void)rag_stage<N_AM_D3, 2, int16_t, 1>(Q_AM_D3, 0, X.r2[r], n2, X.r3[r], tid);
    }
    rag_keep(X.r2[0], n2, N_AM_D3 - 1, tid);
    rag_keep(X.r2[1], n2, N_AM_D3 - 1, tid);
    if (tid == 0)
    {
      X.fst[0] = (kind == 1) ? st->am_x1 : st->ssb_x1;
      X.fst[1] = (kind == 1) ? st->am_y1 : st->ssb_y1;
    }
    if (kind == 4)
    {
      if (tid < N_SSB_DELAY - 1) X.r3[0][-(N_SSB_DELAY - 1) + tid] = st->ssb_i[kSsbHist - (N_SSB_DELAY - 1) + tid];
      if (tid < N_SSB_HILBERT - 1) X.r3[1][-(N_SSB_HILBERT - 1) + tid] = st->ssb_q[kSsbHist - (N_SSB_HILBERT - 1) + tid];
      if (tid < kSsbHist)
      {
        X.h8k[0][tid] = st->ssb_i[tid];
        X.h8k[1][tid] = st->ssb_q[tid];
      }
    }
  }
  __syncthreads();
}

// ---- the pipelines -> ChanState, in the streaming kernels' format (on the grid: every commutator is back at 0)
__device__ __forceinline__ void rag_store_chan(const RagCtx &X, ChanState *so, const int kind)
{
  const int tid = X.tid;
  if (kind == 3)
  {
    if (tid == 0)
    {
      so->wb_theta = X.fst[0];
      so->wb_p = X.fst[1];
      so->wb_y = X.fst[2];
    }
    if (tid < kWbS) so->wb_s[tid] = X.s0[-kWbS + tid];
    if (tid < kWbU) so->wb_u[tid] = X.r1[0][-kWbU + tid];
    if (tid < kWbV) so->wb_v[tid] = X.r2[0][-kWbV + tid];
  }
  else if (kind == 2)
  {
    for (int k = tid; k < 2 * kFmTail; k += kRagThreads)
    {
      so->fm_tail[k] = X.hist8[k];
    }
    if (tid < kWbU) so->fm_u[tid] = X.r1[0][-kWbU + tid];
    if (tid < kWbV) so->fm_v[tid] = X.r2[0][-kWbV + tid];
  }
  else if (kind == 1 || kind == 4)
  {
    uint8_t *tail = (kind == 1) ? so->am_tail : so->ssb_tail;
    for (int k = tid; k < 2 * kAmTail; k += kRagThreads)
    {
      tail[k] = X.hist8[k];
    }
    if (tid == 0)
    {
      if (kind == 1)
      {
        so->am_x1 = X.fst[0];
        so->am_y1 = X.fst[1];
      }
      else
      {
        so->ssb_x1 = X.fst[0];
        so->ssb_y1 = X.fst[1];
      }
    }
    if (kind == 4 && tid < kSsbHist)
    {
      so->ssb_i[tid] = X.h8k[0][tid];
      so->ssb_q[tid] = X.h8k[1][tid];
    }
  }
}

// ---- RagState <-> the pipelines
template <typename T, int STRIDE>
__device__ __forceinline__ void rag_get(const RagQ15 &q, T *x, const int L, const int tid)
{
  if (tid < L)
  {
    x[(-L + tid) * STRIDE] = (T)q.tail[tid];
  }
}
template <typename T, int STRIDE>
__device__ __forceinline__ void rag_put(RagQ15 &q, const T *x, const int L, const int phase, const int tid)
{
  if (tid < L)
  {
    q.tail[tid] = (int16_t)x[(-L + tid) * STRIDE];
  }
  if (tid == 0)
  {
    q.phase = phase;
  }
}

__device__ __forceinline__ void rag_load_rag(RagCtx &X, const RagState *rg, const int kind)
{
  const int tid = X.tid;
  X.ph[0] = X.ph[1] = X.ph[2] = 0;
  if (kind == 3)
  {
    if (tid == 0)
    {
      X.fst[0] = rg->wb.theta;
      X.fst[1] = rg->wb.p;
      X.fst[2] = rg->wb.y;
    }
    rag_get<int16_t, 1>(rg->wb.d1, X.s0, N_WBFM_D1 - 1, tid);
    rag_get<int16_t, 1>(rg->wb.d2, X.r1[0], N_POST_D12 - 1, tid);
    rag_get<int16_t, 1>(rg->wb.d3, X.r2[0], N_AUDIO_D40 - 1, tid);
    X.ph[0] = rg->wb.d1.phase;
    X.ph[1] = rg->wb.d2.phase;
    X.ph[2] = rg->wb.d3.phase;
  }
  else if (kind == 2)
  {
    if (tid < 4)
    {
      X.F[tid] = rg->fm.th[tid];
    }
    rag_get<int8_t, 2>(rg->fm.ti, X.mix, N_FM_TUNER_D32 - 1, tid);
    rag_get<int8_t, 2>(rg->fm.tq, X.mix + 1, N_FM_TUNER_D32 - 1, tid);
    rag_get<int16_t, 1>(rg->fm.d2, X.r1[0], N_POST_D12 - 1, tid);
    rag_get<int16_t, 1>(rg->fm.d3, X.r2[0], N_AUDIO_D40 - 1, tid);
    X.ph[0] = rg->fm.ti.phase;
    X.ph[1] = rg->fm.d2.phase;
    X.ph[2] = rg->fm.d3.phase;
  }
  else if (kind == 1 || kind == 4)
  {
    const RagAs &a = (kind == 1) ? rg->am : rg->ssb;
    if (tid == 0)
    {
      X.fst[0] = a.x1;
      X.fst[1] = a.y1;
    }
    for (int r = 0; r < 2; r++)
    {
      rag_get<int8_t, 2>(a.s[r][0], X.mix + r, N_AM_D1 - 1, tid);
      rag_get<int16_t, 1>(a.s[r][1], X.r1[r], N_AM_D2 - 1, tid);
      rag_get<int16_t, 1>(a.s[r][2], X.r2[r], N_AM_D3 - 1, tid);
    }
    if (kind == 4)
    {
      rag_get<int16_t, 1>(a.delay, X.r3[0], N_SSB_DELAY - 1, tid);
      rag_get<int16_t, 1>(a.hilbert, X.r3[1], N_SSB_HILBERT - 1, tid);
    }
    X.ph[0] = a.s[0][0].phase;
    X.ph[1] = a.s[0][1].phase;
    X.ph[2] = a.s[0][2].phase;
  }
  __syncthreads();
}

__device__ __forceinline__ void rag_store_rag(const RagCtx &X, RagState *rg, const int kind)
{
  const int tid = X.tid;
  if (kind == 3)
  {
    if (tid == 0)
    {
      rg->wb.theta = X.fst[0];
      rg->wb.p = X.fst[1];
      rg->wb.y = X.fst[2];
    }
    rag_put<int16_t, 1>(rg->wb.d1, X.s0, N_WBFM_D1 - 1, X.ph[0], tid);
    rag_put<int16_t, 1>(rg->wb.d2, X.r1[0], N_POST_D12 - 1, X.ph[1], tid);
    rag_put<int16_t, 1>(rg->wb.d3, X.r2[0], N_AUDIO_D40 - 1, X.ph[2], tid);
  }
  else if (kind == 2)
  {
    if (tid < 4)
    {
      rg->fm.th[tid] = X.F[tid];
    }
    rag_put<int8_t, 2>(rg->fm.ti, X.mix, N_FM_TUNER_D32 - 1, X.ph[0], tid);
    rag_put<int8_t, 2>(rg->fm.tq, X.mix + 1, N_FM_TUNER_D32 - 1, X.ph[0], tid);
    rag_put<int16_t, 1>(rg->fm.d2, X.r1[0], N_POST_D12 - 1, X.ph[1], tid);
    rag_put<int16_t, 1>(rg->fm.d3, X.r2[0], N_AUDIO_D40 - 1, X.ph[2], tid);
  }
  else if (kind == 1 || kind == 4)
  {
    RagAs &a = (kind == 1) ? rg->am : rg->ssb;
    if (tid == 0)
    {
      a.x1 = X.fst[0];
      a.y1 = X.fst[1];
    }
    for (int r = 0; r < 2; r++)
    {
      rag_put<int8_t, 2>(a.s[r][0], X.mix + r, N_AM_D1 - 1, X.ph[0], tid);
      rag_put<int16_t, 1>(a.s[r][1], X.r1[r], N_AM_D2 - 1, X.ph[1], tid);
      rag_put<int16_t, 1>(a.s[r][2], X.r2[r], N_AM_D3 - 1, X.ph[2], tid);
    }
    if (kind == 4)
    {
      rag_put<int16_t, 1>(a.delay, X.r3[0], N_SSB_DELAY - 1, 0, tid);
      rag_put<int16_t, 1>(a.hilbert, X.r3[1], N_SSB_HILBERT - 1, 0, tid);
    }
  }
}

// on the grid: the FIR demodulators' input tail follows the stream (the last H samples it consumed, offset binary)
__device__ __forceinline__ void rag_follow_tail(RagCtx &X, const int H, const int n)
{
  uint8_t v[2] = {0, 0};
  for (int j = 0; j < 2; j++)
  {
    const int k = X.tid + j * kRagThreads;
    if (k < 2 * H)
    {
      const int idx = 2 * n - 2 * H + k;
      v[j] = (idx >= 0) ? (uint8_t)((uint8_t)X.mix[idx] ^ 0x80u) : X.hist8[2 * H + idx];
    }
  }
  __syncthreads();
  for (int j = 0; j < 2; j++)
  {
    const int k = X.tid + j * kRagThreads;
    if (k < 2 * H)
    {
      X.hist8[k] = v[j];
    }
  }
}

// ---- WbFmDemodulator::acceptIqData (WbFmDemodulator.cc:341-356) on X.mix[0 .. 2n): returns the PCM count
__device__ __forceinline__ int rag_wbfm(RagCtx &X, const int n, const float gain, int16_t *pcm)
{
  const int tid = X.tid;
  // demodulateSignal (:381-439): theta from the table, all samples at once
  float kgain = gain / 75000.0f;
  kgain = kgain * 32767.0f;
  for (int i = tid; i < n; i += kRagThreads)
  {
    X.F[i] = rag_theta(X, X.mix[2 * i], X.mix[2 * i + 1]);
  }
  __syncthreads();
  // d = wrap(theta - theta_prev); x = K d; the FIR half of the de-emphasis filter v = b0 x + b1 x[n-1] (b1 == b0;
  // IirFilter.cc:161-176).  In place: a thread owns 16 consecutive samples and holds the 18 thetas they need.
  {
    const int i0 = tid * 16;
    float th[18];
#pragma unroll
    for (int k = 0; k < 18; k++)
    {
      const int idx = i0 - 2 + k;
      th[k] = (idx >= 0 && idx < n) ? X.F[idx] : ((idx == -1) ? X.fst[0] : 0.0f);
    }
    const float pcar = X.fst[1];
    __syncthreads();
    if (i0 < n)
    {
      float pprev = (i0 == 0) ? pcar : DEEMPH_B0 * (kgain * wrap_pi(th[1] - th[0]));
#pragma unroll
      for (int k = 0; k < 16; k++)
      {
        const int i = i0 + k;
        if (i < n)
        {
          const float x = kgain * wrap_pi(th[k + 2] - th[k + 1]);
          const float p = DEEMPH_B0 * x;
          X.F[i] = p + pprev;
          pprev = p;
          if (i == n - 1)
          {
            X.fst[0] = th[k + 2];
            X.fst[1] = p;
          }
        }
      }
    }
  }
  __syncthreads();
  // the recursive half: r = a1 y; y = v - r, two rounded operations per sample, in order, on one lane
  if (tid == 0)
  {
    const float a1 = DEEMPH_A1;
    float y = X.fst[2];
    int i = 0;
    for (; i + 4 <= n; i += 4)
    {
      float4 v = *reinterpret_cast<const float4 *>(X.F + i);
      float r = a1 * y;
      y = v.x - r; v.x = y;
      r = a1 * y;
      y = v.y - r; v.y = y;
      r = a1 * y;
      y = v.z - r; v.z = y;
      r = a1 * y;
      y = v.w - r; v.w = y;
      *reinterpret_cast<float4 *>(X.F + i) = v;
    }
    for (; i < n; i++)
    {
      const float r = a1 * y;
      y = X.F[i] - r;
      X.F[i] = y;
    }
    X.fst[2] = y;
  }
  __syncthreads();
  // createPcmData (:460-500): (int16_t)y, D(8,4), D(12,4), D(40,2)
  for (int i = tid; i < n; i += kRagThreads)
  {
    X.s0[i] = (int16_t)f2i16(X.F[i]);
  }
  __syncthreads();
  const int n1 = rag_stage<N_WBFM_D1, 4, int16_t, 1>(Q_WBFM_D1, X.ph[0], X.s0, n, X.r1[0], tid);
  rag_keep(X.s0, n, N_WBFM_D1 - 1, tid);
  X.ph[0] = (X.ph[0] + n) & 3;
  const int n2 = rag_stage<N_POST_D12, 4, int16_t, 1>(Q_POST_D12, X.ph[1], X.r1[0], n1, X.r2[0], tid);
  rag_keep(X.r1[0], n1, N_POST_D12 - 1, tid);
  X.ph[1] = (X.ph[1] + n1) & 3;
  const int n3 = rag_stage<N_AUDIO_D40, 2, int16_t, 1>(Q_AUDIO_D40, X.ph[2], X.r2[0], n2, pcm, tid);
  rag_keep(X.r2[0], n2, N_AUDIO_D40 - 1, tid);
  X.ph[2] = (X.ph[2] + n2) & 1;
  return n3;
}

// ---- FmDemodulator::acceptIqData (FmDemodulator.cc:395-585)
__device__ __forceinline__ int rag_fm(RagCtx &X, const int n, const float gain, int16_t *pcm)
{
  const int tid = X.tid;
  // reduceSampleRate (:395-442): the tuner D(32,4) on both rails
  const int n1 = rag_stage<N_FM_TUNER_D32, 4, int8_t, 2>(Q_FM_TUNER_D32, X.ph[0], X.mix, n, X.r1[0], tid);
  (void)rag_stage<N_FM_TUNER_D32, 4, int8_t, 2>(Q_FM_TUNER_D32, X.ph[0], X.mix + 1, n, X.r1[1], tid);
  rag_keep(X.mix, 2 * n, 2 * (N_FM_TUNER_D32 - 1), tid);
  X.ph[0] = (X.ph[0] + n) & 3;
  // demodulateSignal (:479-529): theta of the LOW BYTES of the tuner's outputs
  for (int k = tid; k < n1; k += kRagThreads)
  {
    X.F[4 + k] = rag_theta(X, X.r1[0][k], X.r1[1][k]);
  }
  __syncthreads();
  // the differentiator {-1/16, 0, 1, 0, -1, 0, 1/16} whose outer taps are integer-division zeros (:116-125):
  // d = theta[k-2] - theta[k-4]; wrap; gain; (int16_t) (:567) -- over the tuner's I outputs, which are dead now
  float kgain = gain / 15000.0f;
  kgain = kgain * 32767.0f;
  for (int k = tid; k < n1; k += kRagThreads)
  {
    const float d = wrap_pi(X.F[k + 2] - X.F[k]);
    X.r1[0][k] = (int16_t)f2i16(kgain * d);
  }
  rag_keep(X.F + 4, n1, 4, tid);
  const int n2 = rag_stage<N_POST_D12, 4, int16_t, 1>(Q_POST_D12, X.ph[1], X.r1[0], n1, X.r2[0], tid);
  rag_keep(X.r1[0], n1, N_POST_D12 - 1, tid);
  X.ph[1] = (X.ph[1] + n1) & 3;
  const int n3 = rag_stage<N_AUDIO_D40, 2, int16_t, 1>(Q_AUDIO_D40, X.ph[2], X.r2[0], n2, pcm, tid);
  rag_keep(X.r2[0], n2, N_AUDIO_D40 - 1, tid);
  X.ph[2] = (X.ph[2] + n2) & 1;
  return n3;
}

// ---- AmDemodulator::acceptIqData (AmDemodulator.cc:339-504) / SsbDemodulator::acceptIqData (SsbDemodulator.cc:462-598)
__device__ __forceinline__ int rag_amssb(RagCtx &X, const int n, const bool ssb, const bool lsb, const float gain,
                                         int16_t *pcm, const bool follow8k)
{
  const int tid = X.tid;
  int n1 = 0, n2 = 0, n3 = 0;
  for (int r = 0; r < 2; r++)
  {
    n1 = rag_stage<N_AM_D1, 4, int8_t, 2>(Q_AM_D1, X.ph[0], X.mix + r, n, X.r1[r], tid);
  }
  rag_keep(X.mix, 2 * n, 2 * (N_AM_D1 - 1), tid);
  X.ph[0] = (X.ph[0] + n) & 3;
  for (int r = 0; r < 2; r++)
  {
    n2 = rag_stage<N_AM_D2, 4, int16_t, 1>(Q_AM_D2, X.ph[1], X.r1[r], n1, X.r2[r], tid);
  }
  rag_keep(X.r1[0], n1, N_AM_D2 - 1, tid);
  rag_keep(X.r1[1], n1, N_AM_D2 - 1, tid);
  X.ph[1] = (X.ph[1] + n1) & 3;
  for (int r = 0; r < 2; r++)
  {
    n3 = rag_stage<N_AM_D3, 2, int16_t, 1>(Q_AM_D3, X.ph[2], X.r2[r], n2, X.r3[r], tid);
  }
  rag_keep(X.r2[0], n2, N_AM_D3 - 1, tid);
  rag_keep(X.r2[1], n2, N_AM_D3 - 1, tid);
  X.ph[2] = (X.ph[2] + n2) & 1;
  if (ssb)
  {
    if (follow8k)
    {
      // ChanState::ssb_i / ssb_q: the last 32 samples of the 8 kS/s rails
      int16_t v[2] = {0, 0};
      if (tid < kSsbHist)
      {
        for (int r = 0; r < 2; r++)
        {
          const int idx = n3 - kSsbHist + tid;
          v[r] = (idx >= 0) ? X.r3[r][idx] : X.h8k[r][kSsbHist + idx];
        }
      }
      __syncthreads();
      if (tid < kSsbHist)
      {
        X.h8k[0][tid] = v[0];
        X.h8k[1][tid] = v[1];
      }
    }
    // the delay line (Q15 tap 1.0 narrows to -32768: a NEGATING delay of 15) and the 31-tap Hilbert transformer
    (void)rag_stage<N_SSB_DELAY, 1, int16_t, 1>(Q_SSB_DELAY, 0, X.r3[0], n3, X.r4[0], tid);
    (void)rag_stage<N_SSB_HILBERT, 1, int16_t, 1>(Q_SSB_HILBERT, 0, X.r3[1], n3, X.r4[1], tid);
    rag_keep(X.r3[0], n3, N_SSB_DELAY - 1, tid);
    rag_keep(X.r3[1], n3, N_SSB_HILBERT - 1, tid);
  }
  // envelope / I -+ Q, then the dc-removal filter b = {1, -1}, a = {-0.95} (IirFilter.cc:161-176) and the gain: in order
  if (tid == 0)
  {
    float x1 = X.fst[0], y1 = X.fst[1];
    const float a1 = DCREM_A1;
    for (int i = 0; i < n3; i++)
    {
      float x;
      if (ssb)
      {
        const int id = X.r4[0][i], qh = X.r4[1][i];
        x = (float)(lsb ? id - qh : id + qh);                   // SsbDemodulator.cc:580-590: int arithmetic, then the cast
      }
      else
      {
        // AmDemodulator.cc:445-458: int16 magnitudes, max + min / 2
        const int16_t im = (int16_t)abs((int)X.r3[0][i]);
        const int16_t qm = (int16_t)abs((int)X.r3[1][i]);
        const int16_t mag = (im > qm) ? (int16_t)(im + (qm >> 1)) : (int16_t)(qm + (im >> 1));
        x = (float)mag;
      }
      const float v = x - x1;                                    // 0 + 1 x + (-1) x[n-1]
      const float r = a1 * y1;
      const float y = v - r;
      x1 = x;
      y1 = y;
      pcm[i] = (int16_t)f2i16(gain * y);
    }
    X.fst[0] = x1;
    X.fst[1] = y1;
  }
  __syncthreads();
  return n3;
}

#define HRFD_RAG_LDS                                                                                        \
  __shared__ __attribute__((aligned(16))) float s_F[kMaxN256 + 16];                                         \
  __shared__ __attribute__((aligned(16))) int8_t s_mix[2 * kRagHead + 2 * kMaxN256 + 16];                   \
  __shared__ __attribute__((aligned(16))) int16_t s_r1[2][kRagHead + kMaxN256 / 4 + 8];                     \
  __shared__ __attribute__((aligned(16))) int16_t s_r2[2][kRagHead + kMaxN256 / 16 + 8];                    \
  __shared__ __attribute__((aligned(16))) int16_t s_r3[2][kRagHead + kMaxN256 / 32 + 8];                    \
  __shared__ __attribute__((aligned(16))) int16_t s_r4[2][kMaxN256 / 32 + 8];                               \
  __shared__ __attribute__((aligned(16))) uint8_t s_hist8[2 * kFmTail];                                     \
  __shared__ int16_t s_h8k[2][kSsbHist];                                                                    \
  __shared__ float s_fst[4];                                                                                \
  RagCtx X;                                                                                                 \
  X.F = s_F;                                                                                                \
  X.mix = s_mix + 2 * kRagHead;                                                                             \
  X.s0 = reinterpret_cast<int16_t *>(s_mix) + kRagHead;                                                     \
  for (int r_ = 0; r_ < 2; r_++)                                                                            \
  {                                                                                                         \
    X.r1[r_] = s_r1[r_] + kRagHead;                                                                         \
    X.r2[r_] = s_r2[r_] + kRagHead;                                                                         \
    X.r3[r_] = s_r3[r_] + kRagHead;                                                                         \
    X.r4[r_] = s_r4[r_];                                                                                    \
    X.h8k[r_] = s_h8k[r_];                                                                                  \
  }                                                                                                         \
  X.hist8 = s_hist8;                                                                                        \
  X.fst = s_fst;                                                                                            \
  X.tid = (int)threadIdx.x;                                                                                 \
  X.ph[0] = X.ph[1] = X.ph[2] = 0;

// ChanState -> RagState for every channel of the handle: the first call that leaves the grid
__global__ __launch_bounds__(kRagThreads) void k_rag_expand(const ChanState *state, RagState *rag, const float *lut,
                                                            const uint32_t n_channels)
{
  HRFD_RAG_LDS
  const uint32_t c = blockIdx.x;
  if (c >= n_channels)
  {
    return;
  }
  X.lut = lut;
  const ChanState *st = state + c;
  RagState *rg = rag + c;
  const int tid = X.tid;
  for (int kind = 1; kind <= 4; kind++)
  {
    rag_zero_heads(X);
    rag_load_chan(X, st, kind);
    rag_store_rag(X, rg, kind);
    __syncthreads();
  }
  if (tid < 32)
  {
    rg->fe_raw[tid] = (tid < 16) ? (int8_t)0 : st->fe_tail[tid - 16];
  }
  if (tid == 0)
  {
    rg->fe_phase = 0u;
    rg->valid = 1u;
  }
}

__global__ __launch_bounds__(kRagThreads) void k_rx_ragged(const RagParams P)
{
  HRFD_RAG_LDS
  __shared__ int8_t s_fe[32];
  __shared__ uint32_t s_mag;
  const uint32_t ci = blockIdx.x;
  if (ci >= P.n_list)
  {
    return;
  }
  const uint32_t c = (P.chan_list != nullptr) ? P.chan_list[ci] : ci;
  const int tid = X.tid;
  X.lut = P.atan2_lut;

  // ---- the launch's bookkeeping (finish_apply's protocol): this kernel never fails a channel by itself; a channel
  //      behind an unrepaired failure of an earlier launch (pipelined submission) started from a stale state and is left alone
  const uint32_t poison = P.chan_poison[c];
  if (tid == 0)
  {
    P.chan_fail[c] = (poison != 0u) ? kFailPoison : 0u;
    if (poison != 0u)
    {
      atomicAdd(&P.counters[kCntFail], 1u);
      atomicAdd(&P.sticky[kCntTotViol], 1u);
    }
    if (c == P.first_channel)
    {
      atomicAdd(&P.sticky[kCntTotLaunch], 1u);
    }
  }
  if (c == P.first_channel && tid < kCntSticky)
  {
    P.next_local[tid] = 0u;
  }
  if (poison != 0u)
  {
    return;
  }

  const ChanCfg cfg = P.cfg[c];
  const int kind = rag_kind(cfg.mode);
  ChanState *st = P.state + c;
  RagState *rg = (P.rag != nullptr) ? P.rag + c : nullptr;
  const bool og = P.offgrid != 0;
  const float gain = (kind == 1) ? cfg.gain_am : (kind == 2) ? cfg.gain_fm : (kind == 3) ? cfg.gain_wbfm : cfg.gain_ssb;

  // ---- state in
  rag_zero_heads(X);
  int fe_p = og ? (int)(rg->fe_phase & 7u) : 0;
  if (tid < 32)
  {
    s_fe[tid] = og ? rg->fe_raw[tid] : ((tid < 16) ? (int8_t)0 : st->fe_tail[tid - 16]);
  }
  uint32_t tracking = st->tracking;
  if (kind != 0)
  {
    if (og)
    {
      rag_load_rag(X, rg, kind);
    }
    else
    {
      rag_load_chan(X, st, kind);
    }
  }
  __syncthreads();
  bool ran = false;

  for (uint32_t b = 0; b < P.n_blocks; b++)
  {
    const int8_t *blk = P.iq + (uint64_t)c * P.ch_stride + (uint64_t)b * P.block_bytes;
    const size_t ounit = (size_t)c * P.out_blocks + P.out_b0 + b;
    int nout;
    bool allowed = true;
    if (!P.src256)
    {
      // ---- IqDataProcessor::reduceSampleRate (:429-500) + upconvertByFsOver4 (:771-815) + SignalDetector (:205-274)
      const int n = (int)(P.block_bytes >> 1);
      nout = (fe_p + n) >> 3;
      if (tid == 0)
      {
        s_mag = 0u;
      }
      __syncthreads();
      uint32_t msum = 0;
      for (int j = tid; j < nout; j += kRagThreads)
      {
        // output j completes at raw sample rj of this call; its receptive field is rj - 14 .. rj per rail
        const int rj = 7 - fe_p + 8 * j;
        int out[2];
#pragma unroll
        for (int r = 0; r < 2; r++)
        {
          int x[15];
#pragma unroll
          for (int k = 0; k < 15; k++)
          {
            const int t = rj - 14 + k;
            x[k] = (t >= 0) ? (int)blk[2 * t + r] : (int)s_fe[32 + 2 * t + r];
          }
          int s1[7];
#pragma unroll
          for (int m = 0; m < 7; m++)
          {
            const uint32_t acc = (1u << 14) + (uint32_t)(Q_HB1[0] * x[2 * m + 2] + Q_HB1[1] * x[2 * m + 1] + Q_HB1[2] * x[2 * m]);
            s1[m] = (int)(int16_t)((int32_t)acc >> 15);
          }
          int s2[3];
#pragma unroll
          for (int k = 0; k < 3; k++)
          {
            const uint32_t acc = (1u << 14) + (uint32_t)(Q_HB2[0] * s1[2 * k + 2] + Q_HB2[1] * s1[2 * k + 1] + Q_HB2[2] * s1[2 * k]);
            s2[k] = (int)(int16_t)((int32_t)acc >> 15);
          }
          const uint32_t acc = (1u << 14) + (uint32_t)(Q_HB3[0] * s2[2] + Q_HB3[1] * s2[1] + Q_HB3[2] * s2[0]);
          out[r] = (int)(int8_t)(int16_t)((int32_t)acc >> 15);  // (int8_t)sample, :458 / :489
        }
        // the rotation by the index WITHIN THE CALL; int8 negation wraps
        int mi = out[0], mq = out[1];
        switch (j & 3)
        {
          case 1: mi = (int)(int8_t)(-out[1]); mq = out[0]; break;
          case 2: mi = (int)(int8_t)(-out[0]); mq = (int)(int8_t)(-out[1]); break;
          case 3: mi = out[1]; mq = (int)(int8_t)(-out[0]); break;
          default: break;
        }
        const uint16_t w = (uint16_t)(((uint32_t)mi & 0xffu) | (((uint32_t)mq & 0xffu) << 8));
        reinterpret_cast<uint16_t *>(X.mix)[j] = w;
        if (P.iq256 != nullptr)
        {
          reinterpret_cast<uint16_t *>(P.iq256 + ounit * (size_t)P.iq256_cap)[j] = w;
        }
        const uint32_t ai = (uint32_t)abs(mi), aq = (uint32_t)abs(mq);   // uint8_t magnitudes: |-128| = 128
        msum += (ai > aq) ? ai + (aq >> 1) : aq + (ai >> 1);
      }
      for (int off = 32; off > 0; off >>= 1)
      {
        msum += __shfl_down(msum, off);
      }
      if ((tid & 63) == 0 && msum != 0u)
      {
        atomicAdd(&s_mag, msum);
      }
      // the front end's memory for the next call: the last 16 raw samples of (history ++ block)
      int8_t nv = 0;
      if (tid < 32)
      {
        const int idx = 2 * n - 32 + tid;
        nv = (idx >= 0) ? blk[idx] : s_fe[32 + idx];
      }
      __syncthreads();
      if (tid < 32)
      {
        s_fe[tid] = nv;
      }
      fe_p = (fe_p + n) & 7;
      const uint32_t total = s_mag;
      // (the reference divides by the call's own count, SignalDetector.cc:255 -- by ZERO for a call that completes no
      //  256 kS/s sample: SIGFPE there, magnitude 0 here)
      const uint32_t mean_mag = (nout > 0) ? total / (uint32_t)nout : 0u;
      int32_t dbfs = P.dbfs[min(mean_mag, 127u)] - 42;
      dbfs = (int32_t)((uint32_t)dbfs - P.gain_db);
      const bool present = dbfs >= cfg.threshold;
      allowed = present || tracking != 0u;                   // Squelch.cc:227-273, SignalTracker.cc:104-146
      tracking = present ? 1u : 0u;
      if (tid == 0)
      {
        P.magnitude[ounit] = mean_mag;
        if (P.allowed != nullptr)
        {
          P.allowed[ounit] = allowed ? 1 : 0;
        }
      }
    }
    else
    {
      // X::acceptIqData(int8_t *, uint32_t) on the mixed 256 kS/s stream: no front end, no squelch
      nout = (int)(P.block_bytes >> 1);
      for (int j = tid; j < nout; j += kRagThreads)
      {
        reinterpret_cast<uint16_t *>(X.mix)[j] = (uint16_t)(((uint32_t)(uint8_t)blk[2 * j]) | ((uint32_t)(uint8_t)blk[2 * j + 1] << 8));
      }
      __syncthreads();
    }
    // ---- the demodulator of the channel's mode (IqDataProcessor.cc:991-1034): only when the gate is open
    int npcm = 0;
    if (allowed && kind != 0)
    {
      int16_t *pcm = P.pcm + ounit * (size_t)P.pcm_cap;
      ran = true;
      if (!og && kind != 3)
      {
        rag_follow_tail(X, (kind == 2) ? kFmTail : kAmTail, nout);
      }
      if (kind == 3)
      {
        npcm = rag_wbfm(X, nout, gain, pcm);
      }
      else if (kind == 2)
      {
        npcm = rag_fm(X, nout, gain, pcm);
      }
      else
      {
        npcm = rag_amssb(X, nout, kind == 4, cfg.lsb != 0, gain, pcm, !og);
      }
    }
    if (tid == 0 && P.n_pcm != nullptr)
    {
      P.n_pcm[ounit] = (uint32_t)npcm;
    }
    __syncthreads();
  }

  // ---- state out
  if (og)
  {
    if (tid < 32)
    {
      rg->fe_raw[tid] = s_fe[tid];
    }
    if (tid == 0)
    {
      rg->fe_phase = (uint32_t)fe_p;
    }
    if (ran)
    {
      rag_store_rag(X, rg, kind);
    }
  }
  else
  {
    if (!P.src256 && tid < 16)
    {
      st->fe_tail[tid] = s_fe[16 + tid];
    }
    if (ran)
    {
      rag_store_chan(X, st, kind);
    }
  }
  if (tid == 0 && !P.src256)
  {
    st->tracking = tracking;
  }
}

} // namespace hrfd
