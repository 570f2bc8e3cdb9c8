// hackrfdiags_amd/csrc/hrfd_api.hip -- host side of the C ABI declared in
// include/hrfd.h.  Owns device memory, per-channel state, streams and launches;
// contains no signal processing of its own and NO CPU fallback: without a HIP
// device every create call fails with HRFD_ENODEV.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "../../include/hrfd.h"
#include "hrfd_device.h"
#include "hrfd_tables.h"

namespace hrfd {
template <int MODE, bool S256, bool ARITH> __global__ void k_rx_wbfm(const RxParams);
__global__ void k_build_atan_corr(const float *, const float *, uint8_t *, uint32_t *);
template <bool TAB> __global__ void k_atan_eval(const uint8_t *, const float *, float *);
template <int MODE, bool S256, bool ARITH> __global__ void k_rx_fir(const RxParams);
template <int MODE> __global__ void k_rx_post(const RxParams);
__global__ void k_rx_ragged(const RagParams);
__global__ void k_rag_expand(const ChanState *, RagState *, const float *, const uint32_t);
__global__ void k_rx_finish(const EpilogueParams);
} // namespace hrfd

using namespace hrfd;

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                    \
  do                                                                                     \
  {                                                                                      \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess)                                                                \
    {                                                                                    \
      return fail(HRFD_ENODEV, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),    \
                  __FILE__, __LINE__);                                                   \
    }                                                                                    \
  } while (0)

extern "C" const char *hrfd_last_error(void) { return g_err; }
extern "C" int hrfd_version(void) { return HRFD_VERSION; }

extern "C" int hrfd_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
  {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// ------------------------------------------------------------------ host-built tables
// Built with the host libm exactly as the reference constructors do, never with
// device intrinsics:
//   atan2 table  WbFmDemodulator.cc:137-148 / FmDemodulator.cc:159-170
//   dBFS table   DbfsCalculator.cc:58-65 (20*log10f(i), truncated)
static void build_atan2(float *out)
{
  for (int x = 0; x < 256; x++)
  {
    for (int y = 0; y < 256; y++)
    {
      const double xa = (double)x - 128;
      const double ya = (double)y - 128;
      out[y * 256 + x] = (float)atan2(ya, xa);
    }
  }
}

// First-quadrant table of the re-split flow kernel (theta_quad, hrfd_rx_kernels.hip): TQ[|q| * 129 + |i|] =
// (float)atan2((double)|q|, (double)|i|) -- the reference's own entry for i, q >= 0 -- with, in its two free top bits,
// the signed correction (ulps) that makes bits(pi_f - t) + fix the reference's entry for i < 0.  Derived from the
// reference table `lut` itself and PROVEN here, entry by entry: the corrections fit two bits, the table is odd in q
// (row q = -128 included), every word is below 2.0.  Returns false when any of that fails on this libm (the library
// then keeps the kernels that do not use this table).
static bool build_atan_quadrant(const float *lut, uint32_t *out)
{
  auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
  const float pi_f = 3.14159274f;
  bool ok = true;
  for (int i = 0; i < kQuadDwords; i++)
  {
    out[i] = 0u;
  }
  for (int aq = 0; aq <= 128; aq++)
  {
    for (int ai = 0; ai <= 128; ai++)
    {
      const float t = (float)atan2((double)aq, (double)ai);
      const uint32_t tb = bits(t);
      ok = ok && tb < 0x40000000u;
      if (aq <= 127 && ai <= 127)
      {
        ok = ok && tb == bits(lut[(aq + 128) * 256 + (ai + 128)]);          // i, q >= 0: the reference's entry itself
      }
      if (aq >= 1 && ai <= 127)
      {
        ok = ok && (tb ^ 0x80000000u) == bits(lut[(128 - aq) * 256 + (ai + 128)]);   // q < 0, i >= 0: the exact negation
      }
      int32_t fix = 0;
      if (ai >= 1)
      {
        // i = -ai: the entry of q = +aq where it exists, else (q = -128) the negated one
        const uint32_t target = (aq <= 127) ? bits(lut[(aq + 128) * 256 + (128 - ai)]) : (bits(lut[0 * 256 + (128 - ai)]) ^ 0x80000000u);
        fix = (int32_t)(target - bits(pi_f - t));
        ok = ok && fix >= -2 && fix <= 1;
        if (aq >= 1 && aq <= 127)
        {
          ok = ok && (target ^ 0x80000000u) == bits(lut[(128 - aq) * 256 + (128 - ai)]);   // q < 0, i < 0: odd in q as well
        }
      }
      out[aq * kQuadRow + ai] = tb | ((uint32_t)(fix & 3) << 30);
    }
  }
  return ok;
}

static void build_dbfs(int32_t *out)
{
  for (int i = 1; i <= 256; i++)
  {
    const float db = 20 * log10f((float)i);
    out[i] = (int32_t)db;
  }
  out[0] = out[1];
}

extern "C" int hrfd_atan2_table(float *out)
{
  if (out == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_atan2_table: NULL");
  }
  build_atan2(out);
  return HRFD_OK;
}

extern "C" int hrfd_dbfs_table(int32_t *out)
{
  if (out == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_dbfs_table: NULL");
  }
  build_dbfs(out);
  return HRFD_OK;
}

extern "C" int hrfd_q15_table(const char *name, int16_t *out, int cap)
{
  if (name == nullptr)
  {
    return 0;
  }
  for (const NamedTable &t : kNamedTables)
  {
    if (strcmp(t.name, name) == 0)
    {
      if (out != nullptr)
      {
        memcpy(out, t.taps, sizeof(int16_t) * (size_t)std::min(cap, t.n));
      }
      return t.n;
    }
  }
  return 0;
}

// ------------------------------------------------------------------ rx handle
#ifndef HRFD_BANK_XCD_ORDER
#define HRFD_BANK_XCD_ORDER 0      /* 1: the mixed bank's WBFM channels on the even XCDs -- MEASURED, NOTHING (profiles/r5_bank_order_ab_NOTHING.txt) */
#endif

struct hrfd_rx
{
  int device = 0;
  int n_cus = 256;                     // compute units of the device
  uint32_t n_channels = 0;
  hipStream_t stream = nullptr;
  hipStream_t last_stream = nullptr;

  std::mutex mu;                       // guards h_cfg / dirty (setters may come from another thread)
  std::vector<ChanCfg> h_cfg;
  bool cfg_dirty = true;
  std::vector<std::pair<uint32_t, int>> pending_resets;   // (channel, mode)

  ChanCfg *d_cfg = nullptr;
  ChanState *d_state = nullptr;
  ChanState *d_state_out = nullptr;
  float *d_lut = nullptr;
  uint8_t *d_atcorr = nullptr;         // arithmetic atan2 (theta_arith): correction bytes, 1/a
  float *d_atinv = nullptr;
  uint8_t *d_atcorr2 = nullptr;        // first-octant table atan2 (theta_tab): correction bytes, T0
  float *d_att0 = nullptr;
  bool tab_ok = false;                 // its corrections fit: k_rx_wbfm_flow may run
  uint32_t *d_atquad = nullptr;        // first-quadrant table with embedded corrections (theta_quad: the re-split WBFM flow kernel)
  bool quad_ok = false;
  bool arith_ok = false;               // corrections fit: k_rx_wbfm computes theta instead of gathering it
  int atan_mode = -1;                  // test hook: -1 auto, 0 force the table gather, 1 require arithmetic
  int32_t *d_dbfs = nullptr;
  uint32_t *d_counters = nullptr;       // [kNumDevCounters] + a second set of the per-launch counters [kCntSticky]
  uint32_t *d_local = nullptr;          // the per-launch counters of the latest launch (set 0 = d_counters, set 1 behind it)
  int parity = 0;
  uint32_t *d_lists = nullptr;         // [10][n_channels] channel ids grouped by mode; list 6: every channel that is not WBFM,
                                       // list 7: the AM and SSB channels, list 9: every channel but those in mode NONE
  uint32_t list_count[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t *d_sub_lists = nullptr;     // the same for a launch over a subset of the channels (replay of failed channels);
                                       // list 6 there: the subset itself
  uint32_t *d_chan = nullptr;          // [4][n_channels]: chan_fail, chan_poison, chan_expired, chan_arrived (EpilogueParams)
  std::vector<uint32_t> h_fail;        // chan_fail of the latest synchronised launch

  // per-call scratch, grown on demand (units = channels * blocks)
  size_t cap_units = 0;
  uint8_t *d_present = nullptr;
  uint32_t *d_magnitude = nullptr;
  float *d_chk_pub = nullptr;
  float *d_chk_spec = nullptr;
  int16_t *d_ssb_iq = nullptr;         // 8 kS/s I/Q of the SSB channels, [units][2][npcm]
  size_t cap_ssb = 0;

  // staging for the host-buffer entry
  size_t cap_iq = 0, cap_pcm = 0, cap_iq256 = 0;
  int8_t *d_iq = nullptr;
  int16_t *d_pcm = nullptr;
  int8_t *d_iq256 = nullptr;
  size_t cap_npcm = 0, cap_allowed = 0, cap_mag_out = 0;
  uint32_t *d_npcm = nullptr;
  uint8_t *d_allowed = nullptr;
  uint32_t *d_mag_out = nullptr;
  uint32_t replays = 0;                // launches redone on the exact path (diagnostic)
  uint32_t total_repairs = 0;          // de-emphasis tiles repaired in place since creation

  // measurement hook: HIP events around the demodulator kernels of a launch
  std::vector<hipEvent_t> ev;           // 2 events per slot; slot = launch index % slots
  uint32_t ev_launches = 0;                               // launches that were bracketed with events so far
  uint32_t ev_every = 1, ev_seen = 0;                     // every ev_every-th launch is bracketed (hrfd_rx_debug_timing_every)

  // test hooks
  unsigned long long *d_dbg = nullptr;  // optional phase stamps (hrfd_rx_debug_stamps)
  size_t dbg_cap = 0;
  int warm = kWarm;
  int stagger = 4;
  int run_len = 0;                     // test hook: blocks per workgroup run of k_rx_wbfm (0 = automatic)
  int use_stream = 2;                  // test hook: 0 = WBFM batches on k_rx_wbfm (runs of blocks, phases in sequence) instead of k_rx_wbfm_flow
  int32_t wbfm_max_threshold = -200;   // the highest squelch threshold among the channels with a demodulator (can a gate close at all?)
  int fir_flow = -1;                   // test hook: AM / SSB / FM batches on the flow kernel's FIR modes: -1 when the bank is large enough, 0 never, 1 always
  int gated_pass = 1;                  // test hook: 0 = no gated second pass on the device (closed gates go back to the host's replay)
  int expire_once = 0;                 // test hook: the next k_rx_wbfm_flow launch treats this wait (1..6) of workgroup 0 as expired
  uint32_t last_counters[kNumCounters] = {0};

  // any block length (hrfd_rx_ragged.hip).  A handle is "on the grid" while every block it was given was a multiple of
  // 512 bytes (inner API: 64): every commutator of the chain is at 0 between calls and ChanState is the whole state.
  // The first block of another length takes it off the grid, for good: RagState per channel, every call on k_rx_ragged.
  bool offgrid = false;
  bool rag_built = false;              // k_rag_expand has run (ChanState -> RagState)
  RagState *d_rag = nullptr;
  uint64_t ragged_launches = 0;        // launches that ran on k_rx_ragged (diagnostic: hrfd_rx_debug_ragged)
};

static int rx_free(hrfd_rx *h)
{
  if (h == nullptr)
  {
    return HRFD_OK;
  }
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  void *ptrs[] = {h->d_cfg, h->d_state, h->d_state_out, h->d_lut, h->d_atcorr, h->d_atinv, h->d_atcorr2, h->d_att0, h->d_atquad, h->d_dbfs, h->d_counters,
                  h->d_lists, h->d_sub_lists, h->d_chan, h->d_present, h->d_magnitude, h->d_chk_pub, h->d_chk_spec,
                  h->d_iq, h->d_pcm, h->d_iq256, h->d_npcm, h->d_allowed, h->d_mag_out, h->d_ssb_iq, h->d_dbg, h->d_rag};
  for (void *p : ptrs)
  {
    if (p) (void)hipFree(p);
  }
  for (hipEvent_t e : h->ev)
  {
    (void)hipEventDestroy(e);
  }
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return HRFD_OK;
}

static ChanCfg default_cfg()
{
  ChanCfg c;
  memset(&c, 0, sizeof(c));
  c.mode = HRFD_MODE_NONE;                               // IqDataProcessor.cc:63
  c.threshold = -200;                                    // IqDataProcessor.cc:121
  c.gain_am = 300;                                       // AmDemodulator.cc:102
  c.gain_fm = (float)(64000 / (2 * M_PI));               // FmDemodulator.cc:173
  c.gain_wbfm = (float)(256000 / (2 * M_PI));            // WbFmDemodulator.cc:151
  c.gain_ssb = 300;                                      // SsbDemodulator.cc ctor
  c.lsb = 1;                                             // SsbDemodulator.cc ctor
  return c;
}

extern "C" int hrfd_rx_create(uint32_t n_channels, int device, hrfd_rx **out)
{
  if (out == nullptr || n_channels == 0)
  {
    return fail(HRFD_EINVAL, "hrfd_rx_create: need n_channels > 0 and a result pointer");
  }
  *out = nullptr;
  if (hrfd_device_count() <= 0)
  {
    return fail(HRFD_ENODEV, "hrfd_rx_create: no HIP device visible (this library has no CPU path)");
  }
  if (device < 0)
  {
    HIP_TRY(hipGetDevice(&device));
  }
  HIP_TRY(hipSetDevice(device));
  hrfd_rx *h = new hrfd_rx;
  h->device = device;
  h->n_channels = n_channels;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
    {
      h->n_cus = cus;
    }
  }
  h->h_cfg.assign(n_channels, default_cfg());
  int rc = HRFD_OK;
  auto alloc = [&](void **p, size_t bytes) -> bool {
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess)
    {
      rc = fail(HRFD_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
      return false;
    }
    return true;
  };
  bool ok = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && alloc((void **)&h->d_cfg, sizeof(ChanCfg) * n_channels);
  ok = ok && alloc((void **)&h->d_state, sizeof(ChanState) * n_channels);
  ok = ok && alloc((void **)&h->d_state_out, sizeof(ChanState) * n_channels);
  ok = ok && alloc((void **)&h->d_lut, sizeof(float) * 65536);
  ok = ok && alloc((void **)&h->d_atcorr, kCorrBytes);
  ok = ok && alloc((void **)&h->d_atinv, sizeof(float) * kInvEntries);
  ok = ok && alloc((void **)&h->d_atcorr2, kCorrBytes);
  ok = ok && alloc((void **)&h->d_att0, sizeof(float) * kCorrBytes);
  ok = ok && alloc((void **)&h->d_atquad, sizeof(uint32_t) * kQuadDwords);
  ok = ok && alloc((void **)&h->d_dbfs, sizeof(int32_t) * 257);
  ok = ok && alloc((void **)&h->d_counters, sizeof(uint32_t) * (kNumDevCounters + kCntSticky));
  ok = ok && alloc((void **)&h->d_lists, sizeof(uint32_t) * 10 * n_channels);
  ok = ok && alloc((void **)&h->d_sub_lists, sizeof(uint32_t) * 10 * n_channels);
  ok = ok && alloc((void **)&h->d_chan, sizeof(uint32_t) * 4 * n_channels);
  if (!ok)
  {
    if (rc == HRFD_OK) rc = fail(HRFD_ENODEV, "hrfd_rx_create: stream creation failed");
    rx_free(h);
    return rc;
  }
  // Zero state == the reference's freshly constructed objects: zero filter
  // pipelines, previousTheta = 0, tracker in NoSignal.  A zero raw/iq256
  // history is exactly equivalent to zero filter state.
  // (offset-binary tails hold 0x80 = value 0.)
  std::vector<ChanState> init(n_channels);
  memset(init.data(), 0, sizeof(ChanState) * n_channels);
  for (auto &s : init)
  {
    memset(s.fm_tail, 0x80, sizeof(s.fm_tail));
    memset(s.am_tail, 0x80, sizeof(s.am_tail));
    memset(s.ssb_tail, 0x80, sizeof(s.ssb_tail));
  }
  std::vector<float> lut(65536);
  int32_t dbfs[257];
  build_atan2(lut.data());
  build_dbfs(dbfs);
  hipError_t e = hipMemcpy(h->d_state, init.data(), sizeof(ChanState) * n_channels, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(h->d_state_out, init.data(), sizeof(ChanState) * n_channels, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(h->d_lut, lut.data(), sizeof(float) * 65536, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(h->d_dbfs, dbfs, sizeof(dbfs), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(h->d_counters, 0, sizeof(uint32_t) * (kNumDevCounters + kCntSticky));
  if (e == hipSuccess) e = hipMemset(h->d_chan, 0, sizeof(uint32_t) * 4 * n_channels);
  h->h_fail.assign(n_channels, 0u);
  h->d_local = h->d_counters;
  // arithmetic atan2: reciprocals from the host (correctly rounded), correction bytes derived on
  // the device from the table just uploaded, with the kernel's own arithmetic (k_build_atan_corr)
  float inv[kInvEntries];
  memset(inv, 0, sizeof(inv));
  for (int a = 1; a <= 128; a++)
  {
    inv[a] = 1.0f / (float)a;
  }
  uint32_t bad = 0;
  if (e == hipSuccess) e = hipMemcpy(h->d_atinv, inv, sizeof(inv), hipMemcpyHostToDevice);
  if (e == hipSuccess)
  {
    hipLaunchKernelGGL(k_build_atan_corr<false>, dim3((kCorrBytes + 255) / 256), dim3(256), 0, 0, h->d_lut, h->d_atinv,
                       h->d_atcorr, h->d_counters + kCntScratch);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(&bad, h->d_counters + kCntScratch, sizeof(bad), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemset(h->d_counters + kCntScratch, 0, sizeof(uint32_t));
  // first-octant table T0[a(a+1)/2 + b] = (float)atan2((double)b, (double)a): the host's libm, the formula of
  // WbFmDemodulator.cc:137-148; its corrections for the other octants are derived the same way
  uint32_t bad2 = 0;
  {
    std::vector<float> t0(kCorrBytes, 0.0f);
    for (int a = 0; a <= 128; a++)
    {
      for (int b = 0; b <= a; b++)
      {
        t0[(size_t)a * (a + 1) / 2 + b] = (float)atan2((double)b, (double)a);
      }
    }
    if (e == hipSuccess) e = hipMemcpy(h->d_att0, t0.data(), sizeof(float) * kCorrBytes, hipMemcpyHostToDevice);
  }
  if (e == hipSuccess)
  {
    hipLaunchKernelGGL(k_build_atan_corr<true>, dim3((kCorrBytes + 255) / 256), dim3(256), 0, 0, h->d_lut, h->d_att0,
                       h->d_atcorr2, h->d_counters + kCntScratch);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(&bad2, h->d_counters + kCntScratch, sizeof(bad2), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemset(h->d_counters + kCntScratch, 0, sizeof(uint32_t));
  {
    std::vector<uint32_t> tq(kQuadDwords);
    h->quad_ok = build_atan_quadrant(lut.data(), tq.data());
    if (e == hipSuccess) e = hipMemcpy(h->d_atquad, tq.data(), sizeof(uint32_t) * kQuadDwords, hipMemcpyHostToDevice);
  }
  if (e != hipSuccess)
  {
    rc = fail(HRFD_ENODEV, "hrfd_rx_create: initial upload failed: %s", hipGetErrorString(e));
    rx_free(h);
    return rc;
  }
  h->arith_ok = (bad == 0) || (HRFD_ABLATE & 512) != 0;   // (512: TIMING EXPERIMENT ONLY)
  h->tab_ok = (bad2 == 0);
  *out = h;
  return HRFD_OK;
}

extern "C" int hrfd_rx_destroy(hrfd_rx *h) { return rx_free(h); }

template <typename F>
static int for_channels(hrfd_rx *h, uint32_t channel, F f)
{
  if (h == nullptr)
  {
    return fail(HRFD_EINVAL, "NULL handle");
  }
  if (channel != HRFD_ALL_CHANNELS && channel >= h->n_channels)
  {
    return fail(HRFD_EINVAL, "channel %u out of range (%u channels)", channel, h->n_channels);
  }
  std::lock_guard<std::mutex> g(h->mu);
  const uint32_t lo = (channel == HRFD_ALL_CHANNELS) ? 0 : channel;
  const uint32_t hi = (channel == HRFD_ALL_CHANNELS) ? h->n_channels : channel + 1;
  for (uint32_t c = lo; c < hi; c++)
  {
    f(c);
  }
  h->cfg_dirty = true;
  return HRFD_OK;
}

extern "C" int hrfd_rx_set_mode(hrfd_rx *h, uint32_t channel, int mode)
{
  if (mode < HRFD_MODE_NONE || mode > HRFD_MODE_USB)
  {
    return fail(HRFD_EINVAL, "hrfd_rx_set_mode: bad mode %d", mode);
  }
  return for_channels(h, channel, [&](uint32_t c) {
    h->h_cfg[c].mode = mode;
    // IqDataProcessor::setDemodulatorMode also selects the SSB sideband (:357-372)
    if (mode == HRFD_MODE_LSB) h->h_cfg[c].lsb = 1;
    if (mode == HRFD_MODE_USB) h->h_cfg[c].lsb = 0;
  });
}

extern "C" int hrfd_rx_set_gain(hrfd_rx *h, uint32_t channel, int mode, float gain)
{
  if (mode < HRFD_MODE_AM || mode > HRFD_MODE_USB)
  {
    return fail(HRFD_EINVAL, "hrfd_rx_set_gain: bad mode %d", mode);
  }
  return for_channels(h, channel, [&](uint32_t c) {
    switch (mode)
    {
      case HRFD_MODE_AM: h->h_cfg[c].gain_am = gain; break;
      case HRFD_MODE_FM: h->h_cfg[c].gain_fm = gain; break;
      case HRFD_MODE_WBFM: h->h_cfg[c].gain_wbfm = gain; break;
      default: h->h_cfg[c].gain_ssb = gain; break;
    }
  });
}

extern "C" int hrfd_rx_set_threshold(hrfd_rx *h, uint32_t channel, int32_t threshold)
{
  return for_channels(h, channel, [&](uint32_t c) { h->h_cfg[c].threshold = threshold; });
}

extern "C" int hrfd_rx_reset_demod(hrfd_rx *h, uint32_t channel, int mode)
{
  if (mode < HRFD_MODE_AM || mode > HRFD_MODE_USB)
  {
    return fail(HRFD_EINVAL, "hrfd_rx_reset_demod: bad mode %d", mode);
  }
  return for_channels(h, channel, [&](uint32_t c) { h->pending_resets.push_back({c, mode}); });
}

static int grow(void **p, size_t *cap, size_t need)
{
  if (need <= *cap && *p != nullptr)
  {
    return HRFD_OK;
  }
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  hipError_t e = hipMalloc(p, need);
  if (e != hipSuccess)
  {
    *cap = 0;
    return fail(HRFD_ENOMEM, "hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
  }
  *cap = need;
  return HRFD_OK;
}

// apply queued X::resetDemodulator calls to the device state
static int apply_resets(hrfd_rx *h, hipStream_t s, std::vector<std::pair<uint32_t, int>> &resets)
{
  for (auto &r : resets)
  {
    if (h->rag_built)
    {
      // off the grid the state is RagState: Decimator_int16::resetFilterState (Decimator_int16.cc:131-147) clears the
      // pipeline AND the commutator position
      RagState *g = h->d_rag + r.first;
      switch (r.second)
      {
        case HRFD_MODE_WBFM:
          HIP_TRY(hipMemsetAsync(&g->wb.theta, 0, sizeof(float), s));
          HIP_TRY(hipMemsetAsync(&g->wb.d1, 0, 3 * sizeof(RagQ15), s));
          break;
        case HRFD_MODE_FM:
          HIP_TRY(hipMemsetAsync(&g->fm, 0, sizeof(RagFm), s));
          break;
        case HRFD_MODE_AM:
          HIP_TRY(hipMemsetAsync(&g->am, 0, sizeof(RagAs), s));
          break;
        default:
          HIP_TRY(hipMemsetAsync(&g->ssb, 0, sizeof(RagAs), s));
          break;
      }
      continue;
    }
    ChanState *d = h->d_state + r.first;
    switch (r.second)
    {
      case HRFD_MODE_WBFM:
        // WbFmDemodulator.cc:265-278: the three decimators and previousTheta;
        // the de-emphasis filter (wb_p, wb_y) is left alone.
        HIP_TRY(hipMemsetAsync(&d->wb_theta, 0, sizeof(float), s));
        HIP_TRY(hipMemsetAsync(d->wb_s, 0, sizeof(d->wb_s) + sizeof(d->wb_u) + sizeof(d->wb_v), s));
        break;
      case HRFD_MODE_FM:
        HIP_TRY(hipMemsetAsync(d->fm_tail, 0x80, sizeof(d->fm_tail), s));
        HIP_TRY(hipMemsetAsync(d->fm_u, 0, sizeof(d->fm_u) + sizeof(d->fm_v), s));
        break;
      case HRFD_MODE_AM:
        HIP_TRY(hipMemsetAsync(d->am_tail, 0x80, sizeof(d->am_tail), s));
        HIP_TRY(hipMemsetAsync(&d->am_x1, 0, 2 * sizeof(float), s));
        break;
      default:
        HIP_TRY(hipMemsetAsync(d->ssb_tail, 0x80, sizeof(d->ssb_tail), s));
        HIP_TRY(hipMemsetAsync(&d->ssb_x1, 0, 2 * sizeof(float) + sizeof(d->ssb_i) + sizeof(d->ssb_q), s));
        break;
    }
  }
  resets.clear();
  return HRFD_OK;
}

struct LaunchOpts
{
  uint32_t out_blocks;     // layout [C][out_blocks] of the caller's output buffers
  uint32_t out_b0;         // first block of that layout this launch fills
  int serial;              // exact one-lane de-emphasis (replay path)
  int src256;              // input is the 256 kS/s mixed stream (hrfd_demod_*)
  const std::vector<uint32_t> *subset = nullptr;   // launch for these channels only (ascending ids), nullptr = all
};

static int rx_launch(hrfd_rx *h, const int8_t *d_iq, uint64_t channel_stride, uint32_t block_bytes,
                     uint32_t n_blocks, uint32_t gain_db, int16_t *d_pcm, uint32_t *d_n_pcm,
                     uint32_t *d_magnitude, uint8_t *d_allowed, int8_t *d_iq256, hipStream_t s,
                     const LaunchOpts &opt)
{
  if (h == nullptr || d_iq == nullptr || d_pcm == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_rx_process: NULL handle or buffer");
  }
  // Lengths.  The reference takes any byteCount (IqDataProcessor.cc:926, DataConsumer.cc:229-241: short transfers are
  // passed on); what it cannot take is refused here: more than its fixed arrays hold (DataConsumer clips to 262144
  // before the call, DataConsumer.cc:229-233; the demodulators' members hold 32768 bytes) and odd counts (its Q loop
  // then reads bufferPtr[byteCount], IqDataProcessor.cc:474: the caller rounds up, as hrfd_shim.cc does).
  const uint32_t max_bytes = opt.src256 ? 32768u : HRFD_BLOCK_BYTES;
  if (block_bytes == 0 || (block_bytes & 1u) != 0 || block_bytes > max_bytes)
  {
    return fail(HRFD_EINVAL, "%s must be even, > 0 and <= %u (got %u)", opt.src256 ? "bytes_per_channel" : "block_bytes",
                max_bytes, block_bytes);
  }
  // the streaming kernels take whole 1 KiB chunks (inner API: 128 bytes) on a handle that never left the grid
  const bool ragged = h->offgrid || (block_bytes % (opt.src256 ? 128u : 1024u)) != 0;
  if (n_blocks == 0 || opt.out_b0 + n_blocks > opt.out_blocks)
  {
    return fail(HRFD_EINVAL, "bad block count");
  }
  if (channel_stride < (uint64_t)block_bytes * n_blocks)
  {
    return fail(HRFD_EINVAL, "channel_stride smaller than n_blocks*block_bytes");
  }
  if ((uint64_t)block_bytes * n_blocks > 0x7fffffffull)
  {
    // the kernels address a channel's input through a 32-bit buffer descriptor (num_records, byte offsets)
    return fail(HRFD_EINVAL, "n_blocks*block_bytes = %llu exceeds 2^31 - 1 bytes per channel and call",
                (unsigned long long)block_bytes * n_blocks);
  }
  HIP_TRY(hipSetDevice(h->device));

  const uint32_t n256 = opt.src256 ? block_bytes / 2 : block_bytes / 16;
  const uint32_t halo_unit = opt.src256 ? 2u : 16u;   // input bytes per 256 kS/s sample
  // De-emphasis tiles of kTile samples end at n256.  A lane starts warm_tiles tiles early from a
  // seed summed over seed_terms tiles, so in a block that has to re-derive its history (the first
  // block of a workgroup's run when b > 0) the first `sac` tiles cannot be started properly: they
  // are sacrificial, and tile `sac` must begin at or before the cross-block check position
  // -(kNeedHist + 1), the first sample the integer stages' history is built from.
  const int warm_tiles = (h->warm >= kWarm) ? kWarmTiles : std::min(kWarmTiles, h->warm / 128);
  const int seed_terms = (h->warm >= kWarm) ? kSeedTerms : 0;
  const int sac = warm_tiles + seed_terms;
  const int ntiles = ((int)n256 + kNeedHist + 1 + kTile - 1) / kTile + sac;
  const int origin = (int)n256 - ntiles * kTile;
  const int hal = (-origin + 63) / 64 * 64;
  if (!ragged && ntiles > kMaxTiles)
  {
    return fail(HRFD_EINVAL, "internal: %d de-emphasis tiles exceed %d", ntiles, kMaxTiles);
  }
  if (!ragged && hal > kMaxHal)
  {
    return fail(HRFD_EINVAL, "internal: history %d exceeds %d", hal, kMaxHal);
  }
  if (!ragged && n_blocks > 1 && (uint32_t)(hal + 64) * halo_unit > block_bytes)
  {
    return fail(HRFD_EINVAL, "blocks of %u bytes are too short for a multi-block call "
                "(need >= %u); submit them one per call", block_bytes, (uint32_t)(hal + 64) * halo_unit);
  }
  if (opt.serial && n_blocks != 1)
  {
    return fail(HRFD_ESTATE, "internal: serial replay needs n_blocks == 1");
  }

  // configuration snapshot
  uint32_t sub_count[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  std::vector<uint32_t> sub_lists;
  std::vector<std::pair<uint32_t, int>> resets;
  {
    std::lock_guard<std::mutex> g(h->mu);
    resets.swap(h->pending_resets);
    if (h->cfg_dirty)
    {
      std::vector<uint32_t> lists((size_t)10 * h->n_channels);
      uint32_t cnt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (uint32_t c = 0; c < h->n_channels; c++)
      {
        const int m = h->h_cfg[c].mode;
        lists[(size_t)m * h->n_channels + cnt[m]++] = c;
        if (m != HRFD_MODE_WBFM)
        {
          lists[(size_t)6 * h->n_channels + cnt[6]++] = c;
        }
        if (m == HRFD_MODE_AM || m == HRFD_MODE_LSB || m == HRFD_MODE_USB)
        {
          lists[(size_t)7 * h->n_channels + cnt[7]++] = c;
        }
        if (m != HRFD_MODE_NONE)
        {
          lists[(size_t)9 * h->n_channels + cnt[9]++] = c;   // list 9: every channel that has a demodulator (k_rx_flow_bank)
        }
      }
      // List 9 runs as ONE launch, position p on XCD p % 8 (map_unit), a workgroup per channel; the workgroups on the XCDs
      // with odd numbers are 3-5 % slower than the others in most launches (profiles/r5_xcd_swap_experiment.txt) and in the
      // mixed bank the WBFM workgroups end ~10 us behind the FIR kinds'.  -DHRFD_BANK_XCD_ORDER=1 puts the WBFM channels on
      // the even positions: MEASURED AND LEFT OFF -- sixteen WBFM workgroups on an XCD instead of eight run slower by what
      // the placement was to gain (the XCDs' clocks are managed one by one), the bank takes the same time
      // (profiles/r5_bank_order_ab_NOTHING.txt).  Which position a channel has changes nothing it computes.
      {
        std::vector<uint32_t> heavy, light;
        for (uint32_t i = 0; i < cnt[9]; i++)
        {
          const uint32_t c = lists[(size_t)9 * h->n_channels + i];
          (h->h_cfg[c].mode == HRFD_MODE_WBFM ? heavy : light).push_back(c);
        }
        size_t ih = 0, il = 0;
        for (uint32_t p = 0; p < cnt[9] && HRFD_BANK_XCD_ORDER; p++)
        {
          const bool want_heavy = (p & 1u) == 0u;
          const bool take_heavy = (want_heavy && ih < heavy.size()) || il >= light.size();
          lists[(size_t)9 * h->n_channels + p] = take_heavy ? heavy[ih++] : light[il++];
        }
      }
      memcpy(h->list_count, cnt, sizeof(cnt));
      h->wbfm_max_threshold = INT32_MIN;
      for (uint32_t c = 0; c < h->n_channels; c++)
      {
        if (h->h_cfg[c].mode != HRFD_MODE_NONE)
        {
          h->wbfm_max_threshold = std::max(h->wbfm_max_threshold, h->h_cfg[c].threshold);
        }
      }
      // synchronous uploads: the host vectors are only valid under the lock
      HIP_TRY(hipStreamSynchronize(s));
      HIP_TRY(hipMemcpy(h->d_cfg, h->h_cfg.data(), sizeof(ChanCfg) * h->n_channels, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(h->d_lists, lists.data(), sizeof(uint32_t) * lists.size(), hipMemcpyHostToDevice));
      h->cfg_dirty = false;
    }
    if (opt.subset != nullptr)
    {
      // per-mode lists of the subset (list 6: the subset itself); only the modes are read under the lock
      sub_lists.resize((size_t)10 * h->n_channels);
      for (uint32_t c : *opt.subset)
      {
        const int m = h->h_cfg[c].mode;
        sub_lists[(size_t)m * h->n_channels + sub_count[m]++] = c;
        sub_lists[(size_t)6 * h->n_channels + sub_count[6]++] = c;
        if (m == HRFD_MODE_AM || m == HRFD_MODE_LSB || m == HRFD_MODE_USB)
        {
          sub_lists[(size_t)7 * h->n_channels + sub_count[7]++] = c;
        }
      }
    }
  }
  if (opt.subset != nullptr && !opt.subset->empty())
  {
    // (outside the configuration lock: the CLI thread's setters do not wait for this upload)
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpy(h->d_sub_lists, sub_lists.data(), sizeof(uint32_t) * sub_lists.size(), hipMemcpyHostToDevice));
  }
  const uint32_t *const list_count = (opt.subset != nullptr) ? sub_count : h->list_count;
  const uint32_t *const d_lists = (opt.subset != nullptr) ? h->d_sub_lists : h->d_lists;
  if (opt.subset != nullptr && opt.subset->empty())
  {
    return HRFD_OK;
  }
  int rc = apply_resets(h, s, resets);
  if (rc != HRFD_OK)
  {
    return rc;
  }

  // launch-local scratch (present flags, cross-block check values) and the
  // magnitude buffer used when the caller does not want one
  const size_t units = (size_t)h->n_channels * n_blocks;
  const size_t ounits = (size_t)h->n_channels * opt.out_blocks;
  if (std::max(units, ounits) > h->cap_units)
  {
    HIP_TRY(hipStreamSynchronize(s));
    const size_t need = std::max(units, ounits);
    size_t c1 = 0, c2 = 0, c3 = 0, c4 = 0;
    h->cap_units = 0;
    if ((rc = grow((void **)&h->d_present, &c1, need)) != HRFD_OK) return rc;
    if ((rc = grow((void **)&h->d_magnitude, &c2, need * 4)) != HRFD_OK) return rc;
    if ((rc = grow((void **)&h->d_chk_pub, &c3, need * 4)) != HRFD_OK) return rc;
    if ((rc = grow((void **)&h->d_chk_spec, &c4, need * 4)) != HRFD_OK) return rc;
    h->cap_units = need;
  }

  if (list_count[HRFD_MODE_LSB] + list_count[HRFD_MODE_USB] != 0)
  {
    const size_t need = units * (size_t)(n256 / 32) * 2 * sizeof(int16_t);
    if (need > h->cap_ssb)
    {
      HIP_TRY(hipStreamSynchronize(s));
      if ((rc = grow((void **)&h->d_ssb_iq, &h->cap_ssb, need)) != HRFD_OK) return rc;
    }
  }
  // per-launch counters: two sets used alternately, each cleared by the previous launch's k_rx_commit
  h->parity ^= 1;
  uint32_t *const local = h->parity ? h->d_counters + kNumDevCounters : h->d_counters;
  uint32_t *const other = h->parity ? h->d_counters : h->d_counters + kNumDevCounters;
  h->d_local = local;

  RxParams P;
  memset(&P, 0, sizeof(P));
  P.iq = d_iq;
  P.ch_stride = channel_stride;
  P.block_bytes = block_bytes;
  P.n_blocks = n_blocks;
  P.n256 = n256;
  P.ntiles = ntiles;
  P.origin = origin;
  P.hal = hal;
  P.warm_tiles = warm_tiles;
  P.seed_terms = seed_terms;
  P.seed_ct = (float)pow(-(double)DEEMPH_A1, (double)kTile);
  P.serial = opt.serial;
  P.src256 = opt.src256;
  P.stagger = h->stagger & 63;
  P.run_len = 1;
  P.n_runs = n_blocks;
  P.dbg_flags = h->stagger >> 8;
  P.out_blocks = opt.out_blocks;
  P.out_b0 = opt.out_b0;
  P.gain_db = gain_db;
  P.state = h->d_state;
  P.state_out = h->d_state_out;
  P.cfg = h->d_cfg;
  P.pcm = d_pcm;
  P.magnitude = (d_magnitude != nullptr) ? d_magnitude : h->d_magnitude;
  P.present = h->d_present;
  P.iq256 = d_iq256;
  P.ssb_iq = h->d_ssb_iq;
  P.atan2_lut = h->d_lut;
  P.at_corr = h->d_atcorr;
  P.at_inv = h->d_atinv;
  P.at_corr2 = h->d_atcorr2;
  P.at_t0 = h->d_att0;
  P.at_quad = h->d_atquad;
  P.dbfs = h->d_dbfs;
  P.chk_pub = h->d_chk_pub;
  P.chk_spec = h->d_chk_spec;
  P.counters = local;
  P.flow_hal = 1536;           // >= 768 + 64 * (warm_tiles + seed_terms + 1), whole units
  P.flow_seed_ct = (float)pow(-(double)DEEMPH_A1, 64.0);
  P.dbg = nullptr;

  EpilogueParams E;
  memset(&E, 0, sizeof(E));
  E.n_channels = h->n_channels;
  E.n_blocks = n_blocks;
  E.n_pcm_per_block = n256 / 32;
  E.out_blocks = opt.out_blocks;
  E.out_b0 = opt.out_b0;
  E.cfg = h->d_cfg;
  E.state = h->d_state;
  E.state_out = h->d_state_out;
  E.present = h->d_present;
  E.allowed = d_allowed;
  E.n_pcm = d_n_pcm;
  E.chk_pub = h->d_chk_pub;
  E.chk_spec = h->d_chk_spec;
  E.counters = local;
  E.sticky = h->d_counters;
  E.next_local = other;
  E.chan_list = nullptr;
  E.first_channel = (opt.subset != nullptr) ? opt.subset->front() : 0u;
  E.chan_fail = h->d_chan;
  E.chan_poison = h->d_chan + h->n_channels;
  E.chan_expired = h->d_chan + 2 * (size_t)h->n_channels;
  E.chan_arrived = h->d_chan + 3 * (size_t)h->n_channels;
  P.fin = E;
  P.self_finish = 0;
  P.sticky = h->d_counters;

  // (an event record is a packet of its own on the queue, ~3 us each: bracketing EVERY launch of a back-to-back
  //  sequence puts ~6 us of gap between kernels that otherwise follow each other without any -- measured, 256 x 16:
  //  0.2237 ms per step with the events, 0.2166 without; hrfd_rx_debug_timing_every samples instead)
  const size_t ev_slots = (h->ev.size() / 2 != 0 && (h->ev_seen++ % h->ev_every) == 0) ? h->ev.size() / 2 : 0;
  const size_t ev_slot = ev_slots ? (h->ev_launches % ev_slots) : 0;
  if (ev_slots)
  {
    HIP_TRY(hipEventRecord(h->ev[2 * ev_slot], s));
  }
  if (ragged)
  {
    // ---------------------------------------------------------------- any block length: k_rx_ragged
    // One workgroup per channel, the call's blocks in order, every stage with its commutator position: exact, no
    // speculation, every channel commits.  A length that is not a whole number of PCM samples (512 bytes; inner API
    // 64) takes the handle off the grid for good: its state moves from ChanState to RagState (k_rag_expand, once).
    if ((block_bytes % (opt.src256 ? 64u : 512u)) != 0)
    {
      h->offgrid = true;
    }
    if (h->offgrid && !h->rag_built)
    {
      if (h->d_rag == nullptr)
      {
        HIP_TRY(hipStreamSynchronize(s));
        hipError_t e = hipMalloc((void **)&h->d_rag, sizeof(RagState) * h->n_channels);
        if (e != hipSuccess)
        {
          h->d_rag = nullptr;
          return fail(HRFD_ENOMEM, "hipMalloc(%zu) failed: %s", sizeof(RagState) * h->n_channels, hipGetErrorString(e));
        }
      }
      HIP_TRY(hipMemsetAsync(h->d_rag, 0, sizeof(RagState) * h->n_channels, s));
      hipLaunchKernelGGL(k_rag_expand, dim3(h->n_channels), dim3(kRagThreads), 0, s, h->d_state, h->d_rag, h->d_lut, h->n_channels);
      HIP_TRY(hipGetLastError());
      h->rag_built = true;
    }
    RagParams R;
    memset(&R, 0, sizeof(R));
    R.iq = d_iq;
    R.ch_stride = channel_stride;
    R.block_bytes = block_bytes;
    R.n_blocks = n_blocks;
    R.src256 = opt.src256;
    R.offgrid = h->offgrid ? 1 : 0;
    R.pcm_cap = opt.src256 ? (block_bytes + 63u) / 64u : (block_bytes + 511u) / 512u;
    R.iq256_cap = 2u * ((block_bytes / 2u + 7u) / 8u);
    R.out_blocks = opt.out_blocks;
    R.out_b0 = opt.out_b0;
    R.chan_list = (opt.subset != nullptr) ? d_lists + (size_t)6 * h->n_channels : nullptr;
    R.n_list = (opt.subset != nullptr) ? list_count[6] : h->n_channels;
    R.gain_db = gain_db;
    R.state = h->d_state;
    R.rag = h->d_rag;
    R.cfg = h->d_cfg;
    R.pcm = d_pcm;
    R.n_pcm = d_n_pcm;
    R.magnitude = (d_magnitude != nullptr) ? d_magnitude : h->d_magnitude;
    R.allowed = d_allowed;
    R.iq256 = d_iq256;
    R.atan2_lut = h->d_lut;
    R.dbfs = h->d_dbfs;
    R.counters = local;
    R.sticky = h->d_counters;
    R.next_local = other;
    R.first_channel = E.first_channel;
    R.chan_fail = E.chan_fail;
    R.chan_poison = E.chan_poison;
    hipLaunchKernelGGL(k_rx_ragged, dim3(R.n_list), dim3(kRagThreads), 0, s, R);
    HIP_TRY(hipGetLastError());
    h->ragged_launches++;
    if (ev_slots)
    {
      HIP_TRY(hipEventRecord(h->ev[2 * ev_slot + 1], s));
      h->ev_launches++;
    }
    h->last_stream = s;
    return HRFD_OK;
  }
  // ---------------------------------------------------------------- dispatch
  // Everything goes to the caller's stream, in this order of preference:
  //  1. k_rx_flow_bank: a bank of several kinds (WBFM, FM, AM / SSB) as ONE launch -- one persistent workgroup per
  //     channel, the mode read per workgroup, every channel finished inside (BASELINE config 3);
  //  2. k_rx_wbfm_flow<.., MODE> per kind, the same shape, when there are channels enough of that kind to fill the
  //     chip that way (WBFM: always; BASELINE configs 2 and 4), behind it the gated pass for WBFM channels whose
  //     squelch gates may close;
  //  3. the block kernels (one workgroup per channel-block: k_rx_wbfm, k_rx_fir + k_rx_post) with k_rx_finish behind
  //     them: single-block calls (the reference's cadence), the inner demodulator API, the exact replay of a subset,
  //     block sizes that are not whole units of 512 samples at 256 kS/s, small banks.
  // The flow shapes need whole units of two 4 KiB pieces per block, at most 64 blocks, and the first-octant table.
  P.dbg = nullptr;
  const uint32_t n_wb = list_count[HRFD_MODE_WBFM], n_as = list_count[7], n_fm = list_count[HRFD_MODE_FM];
  const bool batch = n_blocks > 1 && !opt.serial && !opt.src256 && opt.subset == nullptr;
  // (the flow shapes need their tables: the first-octant one with its corrections -- FM, and the round-4 WBFM build --
  //  and the first-quadrant one of the re-split WBFM chain; both are proven against the reference table at create)
  const bool flow_shape = batch && h->use_stream == 2 && h->tab_ok && (HRFD_FLOW_SPLIT == 0 || h->quad_ok) && h->atan_mode != 0 &&
                          (n256 % 512u) == 0 && n256 >= 2048u;
  const bool flow = flow_shape && n_wb != 0;              // the WBFM channels run on the flow kernel
  const bool fir_shape = flow_shape && h->fir_flow != 0 && n_blocks <= 64u;
  const int kinds = (n_wb != 0) + (n_as != 0) + (n_fm != 0);
  const bool bank = fir_shape && kinds >= 2 && h->fir_flow != 2 && (h->fir_flow > 0 || list_count[9] >= 48u);
  const bool as_flow = !bank && fir_shape && n_as != 0 && (h->fir_flow > 0 || n_as >= 48u);
  const bool fm_flow = !bank && fir_shape && n_fm != 0 && (h->fir_flow > 0 || n_fm >= 48u);
  const bool may_close = (int64_t)h->wbfm_max_threshold > -42 - (int64_t)gain_db;   // can a WBFM gate close at all? (see below)

  // k_rx_wbfm_flow / k_rx_flow_bank over a channel list: one run per channel unless the WBFM bank alone is too small
  // to fill the chip with whole-CU workgroups
  auto launch_flow = [&](int list, uint32_t n, int mode) -> int {
    P.chan_list = d_lists + (size_t)list * h->n_channels;
    P.n_list = n;
    const uint32_t groups = 8u * ((n + 7u) / 8u);
    uint32_t run_len = n_blocks;
    if (mode == HRFD_MODE_WBFM)
    {
      // runs of consecutive blocks per workgroup (only a run's first block re-produces the history in front of it):
      // as long as possible while the launch still fills the chip -- up to the 64 blocks a workgroup can finish from LDS
      // (round 5; rounds 2-4 stopped at 16: a 64-block batch of 256 channels was four runs per channel, each with its own
      // table copy, re-derived history and service tail -- `also.wbfm_256x64` of the bench line)
      run_len = (h->run_len > 0) ? (uint32_t)h->run_len : 64u;
      run_len = std::min(run_len, n_blocks);
      while (h->run_len <= 0 && run_len > 1 && groups * ((n_blocks + run_len - 1) / run_len) < 256u)
      {
        run_len--;
      }
    }
    P.run_len = run_len;
    P.n_runs = (n_blocks + run_len - 1) / run_len;
    const uint32_t grid = groups * P.n_runs;
    P.dbg = (h->d_dbg != nullptr && (size_t)grid * kDbgSlots <= h->dbg_cap) ? h->d_dbg : nullptr;   // (probe builds: one launch per call -- one mode, or the bank)
    P.warm_tiles = std::min(warm_tiles, HRFD_FLOW_WARM_TILES);   // tiles of 64 here (the FIR modes: ring tiles read below a generation)
    P.self_finish = 1;                                     // the last workgroup of a channel finishes it (finish_channel)
    P.dbg_flags |= h->expire_once << 16;
    h->expire_once = 0;
    const bool dump = d_iq256 != nullptr;                   // (`enable iqdump`: the 256 kS/s stream goes out of the stream waves as well)
    if (mode < 0)
    {
      if (dump) hipLaunchKernelGGL((k_rx_flow_bank<HRFD_FLOW_SVC, true>), dim3(grid), dim3(kThreads), 0, s, P);
      else hipLaunchKernelGGL((k_rx_flow_bank<HRFD_FLOW_SVC, false>), dim3(grid), dim3(kThreads), 0, s, P);
    }
    else if (mode == HRFD_MODE_FM)
    {
      if (dump) hipLaunchKernelGGL((k_rx_wbfm_flow<HRFD_FLOW_SVC, false, true, 2>), dim3(grid), dim3(kThreads), 0, s, P);
      else hipLaunchKernelGGL((k_rx_wbfm_flow<HRFD_FLOW_SVC, false, false, 2>), dim3(grid), dim3(kThreads), 0, s, P);
    }
    else if (mode != HRFD_MODE_WBFM)
    {
      if (dump) hipLaunchKernelGGL((k_rx_wbfm_flow<HRFD_FLOW_SVC, false, true, 14>), dim3(grid), dim3(kThreads), 0, s, P);
      else hipLaunchKernelGGL((k_rx_wbfm_flow<HRFD_FLOW_SVC, false, false, 14>), dim3(grid), dim3(kThreads), 0, s, P);
    }
    else if (d_iq256 != nullptr)
    {
      hipLaunchKernelGGL((k_rx_wbfm_flow<HRFD_FLOW_SVC, false, true>), dim3(grid), dim3(kThreads), 0, s, P);
    }
    else
    {
      hipLaunchKernelGGL((k_rx_wbfm_flow<HRFD_FLOW_SVC, false, false>), dim3(grid), dim3(kThreads), 0, s, P);
    }
    P.dbg_flags &= 0xffff;
    P.dbg = nullptr;
    HIP_TRY(hipGetLastError());
    // Squelch (Squelch.cc:227-273, IqDataProcessor.cc:961-1034).  The detector's lowest level is 0 - 42 - gain_db dBFS
    // (DbfsCalculator.cc:111-147): with a threshold at or below it -- the reference's default is -200 -- no gate of
    // the bank can ever close and the batch launch is all there is.  Otherwise the gated pass follows, one launch per
    // kind: its workgroups redo the channels that failed on a closed gate, exactly, and the others leave at once.
    if (h->gated_pass && n_blocks <= 64u && may_close)
    {
      P.run_len = n_blocks;
      P.n_runs = 1;
      auto gated = [&](int glist, uint32_t gn, int gmode) {
        if (gn == 0)
        {
          return;
        }
        P.chan_list = d_lists + (size_t)glist * h->n_channels;
        P.n_list = gn;
        const dim3 gg(8u * ((gn + 7u) / 8u));
        if (gmode == HRFD_MODE_WBFM)
        {
          hipLaunchKernelGGL((k_rx_wbfm_flow<HRFD_FLOW_SVC, true, false>), gg, dim3(kThreads), 0, s, P);
        }
        else if (gmode == HRFD_MODE_FM)
        {
          hipLaunchKernelGGL((k_rx_wbfm_flow<HRFD_FLOW_SVC, true, false, 2>), gg, dim3(kThreads), 0, s, P);
        }
        else
        {
          hipLaunchKernelGGL((k_rx_wbfm_flow<HRFD_FLOW_SVC, true, false, 14>), gg, dim3(kThreads), 0, s, P);
        }
      };
      if (mode < 0 || mode == HRFD_MODE_WBFM) gated(HRFD_MODE_WBFM, n_wb, HRFD_MODE_WBFM);
      if (mode < 0 || mode == HRFD_MODE_FM) gated(HRFD_MODE_FM, n_fm, HRFD_MODE_FM);
      if (mode < 0 || (mode != HRFD_MODE_WBFM && mode != HRFD_MODE_FM && mode >= 0)) gated(7, n_as, HRFD_MODE_AM);
      HIP_TRY(hipGetLastError());
    }
    P.self_finish = 0;
    P.warm_tiles = warm_tiles;
    return HRFD_OK;
  };
  // the block kernels of mode NONE (front end and squelch only) and WBFM: runs of blocks per workgroup
  auto launch_wbfm_blocks = [&](int m) -> int {
    const uint32_t n = list_count[m];
    P.chan_list = d_lists + (size_t)m * h->n_channels;
    P.n_list = n;
    const uint32_t groups = 8u * ((n + 7u) / 8u);
    uint32_t run_len = (h->run_len > 0) ? (uint32_t)h->run_len : 8u;
    run_len = std::min(run_len, n_blocks);
    while (h->run_len <= 0 && run_len > 1 && groups * ((n_blocks + run_len - 1) / run_len) < 512u)
    {
      run_len--;
    }
    if (opt.serial || opt.src256)
    {
      run_len = 1;
    }
    P.run_len = run_len;
    P.n_runs = (n_blocks + run_len - 1) / run_len;
    const uint32_t grid = groups * P.n_runs;
    P.dbg = (h->d_dbg != nullptr && (size_t)grid * kDbgSlots <= h->dbg_cap && m == HRFD_MODE_WBFM) ? h->d_dbg : nullptr;
    if (m == HRFD_MODE_NONE)
    {
      hipLaunchKernelGGL((k_rx_wbfm<0, false, false>), dim3(grid), dim3(kThreads), 0, s, P);
    }
    else if (opt.src256)
    {
      hipLaunchKernelGGL((k_rx_wbfm<3, true, false>), dim3(grid), dim3(kThreads), 0, s, P);
    }
    else if (h->arith_ok && h->atan_mode != 0)
    {
      hipLaunchKernelGGL((k_rx_wbfm<3, false, true>), dim3(grid), dim3(kThreads), 0, s, P);
    }
    else
    {
      hipLaunchKernelGGL((k_rx_wbfm<3, false, false>), dim3(grid), dim3(kThreads), 0, s, P);
    }
    P.dbg = nullptr;
    HIP_TRY(hipGetLastError());
    return HRFD_OK;
  };

  if (bank)
  {
    if ((rc = launch_flow(9, list_count[9], -1)) != HRFD_OK) return rc;
  }
  else
  {
    // AM and SSB: one launch for both kinds (the same three decimators), then their 8 kS/s recurrences
    if (as_flow)
    {
      if ((rc = launch_flow(7, n_as, HRFD_MODE_AM)) != HRFD_OK) return rc;
    }
    else if (n_as != 0)
    {
      P.chan_list = d_lists + (size_t)7 * h->n_channels;
      P.n_list = n_as;
      const uint32_t grid = 8u * ((n_as + 7u) / 8u) * n_blocks;
      if (opt.src256)
      {
        hipLaunchKernelGGL((k_rx_fir<14, true, false>), dim3(grid), dim3(kThreads), 0, s, P);
      }
      else
      {
        hipLaunchKernelGGL((k_rx_fir<14, false, false>), dim3(grid), dim3(kThreads), 0, s, P);
      }
      hipLaunchKernelGGL(k_rx_post<14>, dim3(n_as), dim3(256), 0, s, P);
      HIP_TRY(hipGetLastError());
    }
    if (fm_flow)
    {
      if ((rc = launch_flow(HRFD_MODE_FM, n_fm, HRFD_MODE_FM)) != HRFD_OK) return rc;
    }
    else if (n_fm != 0)
    {
      P.chan_list = d_lists + (size_t)HRFD_MODE_FM * h->n_channels;
      P.n_list = n_fm;
      const uint32_t grid = 8u * ((n_fm + 7u) / 8u) * n_blocks;
      if (opt.src256)
      {
        hipLaunchKernelGGL((k_rx_fir<2, true, false>), dim3(grid), dim3(kThreads), 0, s, P);
      }
      else if (h->arith_ok && h->atan_mode != 0)
      {
        hipLaunchKernelGGL((k_rx_fir<2, false, true>), dim3(grid), dim3(kThreads), 0, s, P);
      }
      else
      {
        hipLaunchKernelGGL((k_rx_fir<2, false, false>), dim3(grid), dim3(kThreads), 0, s, P);
      }
      HIP_TRY(hipGetLastError());
    }
    if (flow)
    {
      if ((rc = launch_flow(HRFD_MODE_WBFM, n_wb, HRFD_MODE_WBFM)) != HRFD_OK) return rc;
    }
    else if (n_wb != 0)
    {
      if ((rc = launch_wbfm_blocks(HRFD_MODE_WBFM)) != HRFD_OK) return rc;
    }
  }
  if (list_count[HRFD_MODE_NONE] != 0)
  {
    if ((rc = launch_wbfm_blocks(HRFD_MODE_NONE)) != HRFD_OK) return rc;
  }
  // the channels that no kernel finished by itself
  auto finish_list = [&](const uint32_t *list, uint32_t n) -> int {
    if (n != 0)
    {
      EpilogueParams G = E;
      G.chan_list = list;
      G.n_channels = n;
      hipLaunchKernelGGL(k_rx_finish, dim3(n), dim3(64), 0, s, G);
      HIP_TRY(hipGetLastError());
    }
    return HRFD_OK;
  };
  if (opt.subset != nullptr)
  {
    if ((rc = finish_list(d_lists + (size_t)6 * h->n_channels, list_count[6])) != HRFD_OK) return rc;   // the subset itself
  }
  else if (!bank && !flow && !as_flow && !fm_flow)
  {
    if ((rc = finish_list(nullptr, h->n_channels)) != HRFD_OK) return rc;                                  // everything, one launch
  }
  else
  {
    for (int m : {HRFD_MODE_NONE, HRFD_MODE_AM, HRFD_MODE_FM, HRFD_MODE_WBFM, HRFD_MODE_LSB, HRFD_MODE_USB})
    {
      const bool self = (m == HRFD_MODE_NONE) ? false : bank || (m == HRFD_MODE_WBFM ? flow : m == HRFD_MODE_FM ? fm_flow : as_flow);
      if (!self)
      {
        if ((rc = finish_list(d_lists + (size_t)m * h->n_channels, list_count[m])) != HRFD_OK) return rc;
      }
    }
  }
  if (ev_slots)
  {
    HIP_TRY(hipEventRecord(h->ev[2 * ev_slot + 1], s));
    h->ev_launches++;
  }
  h->last_stream = s;
  return HRFD_OK;
}

extern "C" int hrfd_rx_process_device(hrfd_rx *h, const int8_t *d_iq, uint64_t channel_stride,
                                      uint32_t block_bytes, uint32_t n_blocks, uint32_t gain_db,
                                      int16_t *d_pcm, uint32_t *d_n_pcm, uint32_t *d_magnitude,
                                      uint8_t *d_signal_allowed, int8_t *d_iq256k_opt, void *stream)
{
  if (h == nullptr)
  {
    return fail(HRFD_EINVAL, "NULL handle");
  }
  hipStream_t s = (stream != nullptr) ? (hipStream_t)stream : h->stream;
  const LaunchOpts opt = {n_blocks, 0, 0, 0};
  return rx_launch(h, d_iq, channel_stride, block_bytes, n_blocks, gain_db, d_pcm, d_n_pcm,
                   d_magnitude, d_signal_allowed, d_iq256k_opt, s, opt);
}

extern "C" int hrfd_rx_sync(hrfd_rx *h, uint32_t *n_violations)
{
  if (h == nullptr)
  {
    return fail(HRFD_EINVAL, "NULL handle");
  }
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = h->last_stream ? h->last_stream : h->stream;
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipMemcpy(h->last_counters, h->d_counters, sizeof(h->last_counters), hipMemcpyDeviceToHost));   // the totals
  HIP_TRY(hipMemcpy(h->last_counters, h->d_local, sizeof(uint32_t) * kCntSticky, hipMemcpyDeviceToHost)); // the latest launch
  h->total_repairs = h->last_counters[kCntTotRepair];
  // channels of the latest launch that did not commit (their own checks failed, or they ran behind an unrepaired failure)
  const uint32_t viol = (h->last_counters[kCntTotLaunch] != 0) ? h->last_counters[kCntFail] : 0u;
  h->last_counters[kCntCommit] = (viol == 0) ? 1u : 0u;  // shown as "all committed" by hrfd_rx_debug_counters
  if (viol != 0)
  {
    // the caller repairs those channels from here (resubmits them block by block, hrfd_rx_failed_channels says
    // which): they may commit again
    HIP_TRY(hipMemcpy(h->h_fail.data(), h->d_chan, sizeof(uint32_t) * h->n_channels, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(h->d_chan + h->n_channels, 0, sizeof(uint32_t) * h->n_channels));
  }
  else
  {
    std::fill(h->h_fail.begin(), h->h_fail.end(), 0u);
  }
  if (n_violations != nullptr)
  {
    *n_violations = viol;
  }
  return HRFD_OK;
}

// Which channels of the launch that hrfd_rx_sync last waited for did not commit: out[c] != 0 (kFail* bits:
// 1 closed gate in a batch, 2 failed time-parallel speculation, 4 behind an unrepaired failure, 8 internal wait expired).
extern "C" int hrfd_rx_failed_channels(hrfd_rx *h, uint8_t *out, uint32_t n)
{
  if (h == nullptr || out == nullptr || n != h->n_channels)
  {
    return fail(HRFD_EINVAL, "hrfd_rx_failed_channels: need a handle and room for n_channels flags");
  }
  for (uint32_t c = 0; c < n; c++)
  {
    out[c] = (uint8_t)h->h_fail[c];
  }
  return HRFD_OK;
}

// Replays `subset` (ascending channel ids) through the exact path: one block per launch (state advances in
// order); a channel whose de-emphasis tiles did not re-synchronise is redone on the one-lane path.  Inputs and
// outputs are the full [n_channels][n_blocks][...] device buffers of the call being repaired.
// (round 6: the blocks [b0, b0 + nb) of the call -- hrfd_rx_process_block repairs a long call chunk by chunk)
static int rx_replay(hrfd_rx *h, const std::vector<uint32_t> &subset, const int8_t *d_iq, uint64_t stride,
                     uint32_t block_bytes, uint32_t n_blocks, uint32_t gain_db, int16_t *d_pcm, uint32_t *d_npcm,
                     uint32_t *d_mag, uint8_t *d_allowed, int8_t *d_iq256, hipStream_t s, bool pcm_is_clear,
                     uint32_t b0 = 0, uint32_t nb = 0)
{
  if (subset.empty())
  {
    return HRFD_OK;
  }
  if (nb == 0)
  {
    nb = n_blocks - b0;
  }
  // squelched units write no PCM: they must read as zeros, not as what a failed batch left there
  const size_t blk_row = (size_t)((block_bytes + 511u) / 512u) * sizeof(int16_t);
  const size_t row = (size_t)n_blocks * blk_row;
  const bool whole_bank = subset.size() == h->n_channels;
  if (pcm_is_clear)
  {
    // (no batch ran over this buffer: the caller's memset still stands)
  }
  else if (whole_bank && nb == n_blocks)
  {
    HIP_TRY(hipMemsetAsync(d_pcm, 0, row * h->n_channels, s));
  }
  else
  {
    for (uint32_t c : subset)
    {
      HIP_TRY(hipMemsetAsync(reinterpret_cast<char *>(d_pcm) + row * c + blk_row * b0, 0, blk_row * nb, s));
    }
  }
  for (uint32_t b = b0; b < b0 + nb; b++)
  {
    // attempt 0: the exact per-block kernel; attempt 1: its one-lane de-emphasis for the channels whose tiles did not
    // re-synchronise.  The whole bank runs on the cached per-mode lists (no subset, no upload: the reference's own
    // cadence of one block per call takes this path on every call).
    bool all = whole_bank, clean = false;
    std::vector<uint32_t> todo;
    if (!all)
    {
      todo = subset;
    }
    for (int attempt = 0; attempt < 2 && !clean; attempt++)
    {
      LaunchOpts opt = {n_blocks, b, attempt, 0};
      opt.subset = all ? nullptr : &todo;
      int rc = rx_launch(h, d_iq + (size_t)b * block_bytes, stride, block_bytes, 1, gain_db, d_pcm, d_npcm, d_mag,
                         d_allowed, d_iq256, s, opt);
      if (rc != HRFD_OK) return rc;
      uint32_t viol = 0;
      if ((rc = hrfd_rx_sync(h, &viol)) != HRFD_OK) return rc;
      if (viol == 0)
      {
        clean = true;
        break;
      }
      h->replays++;
      std::vector<uint32_t> again;
      for (uint32_t c : (all ? subset : todo))
      {
        if (h->h_fail[c] != 0) again.push_back(c);
      }
      todo.swap(again);
      all = false;
    }
    if (!clean)
    {
      return fail(HRFD_ESTATE, "internal: exact replay still reports %zu failed channel(s)", todo.size());
    }
  }
  return HRFD_OK;
}

extern "C" int hrfd_rx_process_block(hrfd_rx *h, const int8_t *iq, uint32_t block_bytes,
                                     uint32_t n_blocks, uint32_t gain_db, int16_t *pcm,
                                     uint32_t *n_pcm, uint32_t *magnitude, uint8_t *signal_allowed,
                                     int8_t *iq256k_opt)
{
  if (h == nullptr || iq == nullptr || pcm == nullptr || n_pcm == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_rx_process_block: NULL argument");
  }
  if (block_bytes == 0 || (block_bytes & 1u) != 0 || block_bytes > HRFD_BLOCK_BYTES || n_blocks == 0)
  {
    return fail(HRFD_EINVAL, "block_bytes must be even, > 0 and <= %u, n_blocks > 0 (got %u, %u)", HRFD_BLOCK_BYTES,
                block_bytes, n_blocks);
  }
  HIP_TRY(hipSetDevice(h->device));
  const uint32_t C = h->n_channels;
  const size_t units = (size_t)C * n_blocks;
  const uint32_t npcm = (block_bytes + 511u) / 512u;        // row lengths: hrfd_rx_pcm_capacity / hrfd_rx_iq256_capacity
  const uint32_t n256b = 2u * ((block_bytes / 2u + 7u) / 8u);
  const size_t iq_bytes = units * block_bytes;
  const size_t pcm_bytes = units * npcm * sizeof(int16_t);
  const size_t iq256_bytes = units * n256b;
  hipStream_t s = h->stream;
  int rc;
  HIP_TRY(hipStreamSynchronize(s));
  if ((rc = grow((void **)&h->d_iq, &h->cap_iq, iq_bytes)) != HRFD_OK) return rc;
  if ((rc = grow((void **)&h->d_pcm, &h->cap_pcm, pcm_bytes)) != HRFD_OK) return rc;
  if (iq256k_opt != nullptr)
  {
    if ((rc = grow((void **)&h->d_iq256, &h->cap_iq256, iq256_bytes)) != HRFD_OK) return rc;
  }
  if ((rc = grow((void **)&h->d_npcm, &h->cap_npcm, units * 4)) != HRFD_OK) return rc;
  if ((rc = grow((void **)&h->d_allowed, &h->cap_allowed, units)) != HRFD_OK) return rc;
  if ((rc = grow((void **)&h->d_mag_out, &h->cap_mag_out, units * 4)) != HRFD_OK) return rc;
  HIP_TRY(hipMemcpyAsync(h->d_iq, iq, iq_bytes, hipMemcpyHostToDevice, s));
  // mode NONE / squelched units produce no PCM: hand back zeros rather than stale bytes
  HIP_TRY(hipMemsetAsync(h->d_pcm, 0, pcm_bytes, s));

  const uint64_t stride = (uint64_t)block_bytes * n_blocks;
  int8_t *d_iq256 = iq256k_opt ? h->d_iq256 : nullptr;
  uint32_t viol = 0;
  if (h->offgrid || (block_bytes % 1024u) != 0)
  {
    // any length: one launch of k_rx_ragged takes the whole call, block by block and exactly (rx_launch)
    if (d_iq256 != nullptr)
    {
      HIP_TRY(hipMemsetAsync(d_iq256, 0, iq256_bytes, s));   // (a row is filled up to the call's own count)
    }
    const LaunchOpts opt = {n_blocks, 0, 0, 0};
    rc = rx_launch(h, h->d_iq, stride, block_bytes, n_blocks, gain_db, h->d_pcm, h->d_npcm, h->d_mag_out, h->d_allowed,
                   d_iq256, s, opt);
    if (rc != HRFD_OK) return rc;
    if ((rc = hrfd_rx_sync(h, &viol)) != HRFD_OK) return rc;
    if (viol != 0)
    {
      return fail(HRFD_ESTATE, "internal: the exact path reported %u uncommitted channel(s)", viol);
    }
  }
  else
  {
  // Round 6: a call of more than 64 blocks runs as CHUNKS of at most 64, one batch launch each (every chunk then has the
  // shapes a 64-block call has: the flow kernels for the FIR modes, and the gated pass on the device behind them -- until
  // round 5 a long call with closing gates went back to the host block by block: +6 ms for 64 channels x 80 blocks).  The
  // chunks follow each other on the stream; the host looks at every chunk's verdict before the next one starts, so a
  // channel that did not commit is replayed over ITS chunk's blocks from the state the chunk in front left.
  const bool batch_ok = (uint32_t)(kMaxHal + 64) * 16u <= block_bytes;
  for (uint32_t b0 = 0; b0 < n_blocks; b0 += 64u)
  {
    const uint32_t nb = std::min(64u, n_blocks - b0);
    std::vector<uint32_t> redo;                            // channels to run on the exact per-block path
    const bool batch_ran = nb > 1 && batch_ok;
    if (batch_ran)
    {
      // the chunk in one launch, blocks of a channel in parallel (speculative)
      const LaunchOpts opt = {n_blocks, b0, 0, 0};
      rc = rx_launch(h, h->d_iq + (size_t)b0 * block_bytes, stride, block_bytes, nb, gain_db, h->d_pcm, h->d_npcm,
                     h->d_mag_out, h->d_allowed, d_iq256, s, opt);
      if (rc != HRFD_OK) return rc;
      if ((rc = hrfd_rx_sync(h, &viol)) != HRFD_OK) return rc;
      for (uint32_t c = 0; c < C && viol != 0; c++)
      {
        if (h->h_fail[c] != 0) redo.push_back(c);
      }
    }
    else
    {
      for (uint32_t c = 0; c < C; c++) redo.push_back(c);
    }
    if ((rc = rx_replay(h, redo, h->d_iq, stride, block_bytes, n_blocks, gain_db, h->d_pcm, h->d_npcm, h->d_mag_out,
                        h->d_allowed, d_iq256, s, !batch_ran, b0, nb)) != HRFD_OK)
    {
      return rc;
    }
  }
  }
  HIP_TRY(hipMemcpyAsync(pcm, h->d_pcm, pcm_bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(n_pcm, h->d_npcm, units * 4, hipMemcpyDeviceToHost, s));
  if (signal_allowed != nullptr)
  {
    HIP_TRY(hipMemcpyAsync(signal_allowed, h->d_allowed, units, hipMemcpyDeviceToHost, s));
  }
  if (magnitude != nullptr)
  {
    HIP_TRY(hipMemcpyAsync(magnitude, h->d_mag_out, units * 4, hipMemcpyDeviceToHost, s));
  }
  if (iq256k_opt != nullptr)
  {
    HIP_TRY(hipMemcpyAsync(iq256k_opt, h->d_iq256, iq256_bytes, hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  return HRFD_OK;
}

// IqDataProcessor::reduceSampleRate as a call of its own (IqDataProcessor.cc:429-500: public in the reference): the
// three half-band stages per rail over one block of every channel, the decimator pipelines advanced, nothing else --
// no squelch, no demodulator.  Here the front end only exists fused with the Fs/4 mixer and the squelch detector, so a
// mode-NONE block runs and the squelch tracker's state is put back afterwards; iq256k receives the MIXED stream
// (upconvertByFsOver4 applied: the caller takes it out again if it wants the reference's decimatedData).
extern "C" int hrfd_rx_reduce_sample_rate(hrfd_rx *h, const int8_t *iq, uint32_t block_bytes, int8_t *iq256k)
{
  if (h == nullptr || iq == nullptr || iq256k == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_rx_reduce_sample_rate: NULL argument");
  }
  HIP_TRY(hipSetDevice(h->device));
  const uint32_t C = h->n_channels;
  std::vector<int> modes(C);
  std::vector<uint32_t> tracking(C), npcm(C);
  std::vector<int16_t> pcm((size_t)C * (block_bytes / 512 + 2));
  {
    std::lock_guard<std::mutex> g(h->mu);
    for (uint32_t c = 0; c < C; c++)
    {
      modes[c] = h->h_cfg[c].mode;
      h->h_cfg[c].mode = HRFD_MODE_NONE;
    }
    h->cfg_dirty = true;
  }
  // (from here on every path puts the modes back)
  int rc = HRFD_OK;
  hipError_t e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess)
  {
    e = hipMemcpy2D(tracking.data(), sizeof(uint32_t), &h->d_state->tracking, sizeof(ChanState), sizeof(uint32_t), C, hipMemcpyDeviceToHost);
  }
  if (e == hipSuccess)
  {
    rc = hrfd_rx_process_block(h, iq, block_bytes, 1, 0, pcm.data(), npcm.data(), nullptr, nullptr, iq256k);
    e = hipMemcpy2D(&h->d_state->tracking, sizeof(ChanState), tracking.data(), sizeof(uint32_t), sizeof(uint32_t), C, hipMemcpyHostToDevice);
  }
  {
    std::lock_guard<std::mutex> g(h->mu);
    for (uint32_t c = 0; c < C; c++)
    {
      h->h_cfg[c].mode = modes[c];
    }
    h->cfg_dirty = true;
  }
  if (e != hipSuccess)
  {
    return fail(HRFD_ENODEV, "hrfd_rx_reduce_sample_rate: %s", hipGetErrorString(e));
  }
  return rc;
}

// Row lengths of the outputs for a block length, and what the front end holds back between calls.
extern "C" uint32_t hrfd_rx_pcm_capacity(uint32_t block_bytes) { return (block_bytes + 511u) / 512u; }
extern "C" uint32_t hrfd_rx_iq256_capacity(uint32_t block_bytes) { return 2u * ((block_bytes / 2u + 7u) / 8u); }
extern "C" uint32_t hrfd_demod_pcm_capacity(uint32_t bytes_per_channel) { return (bytes_per_channel + 63u) / 64u; }

extern "C" int hrfd_rx_pending_samples(hrfd_rx *h, uint32_t *pending)
{
  if (h == nullptr || pending == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_rx_pending_samples: NULL argument");
  }
  *pending = 0;
  if (!h->rag_built)
  {
    return HRFD_OK;                                        // on the grid: every call ended on a whole 256 kS/s sample
  }
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = h->last_stream ? h->last_stream : h->stream;
  HIP_TRY(hipStreamSynchronize(s));
  uint32_t p = 0;
  HIP_TRY(hipMemcpy(&p, &h->d_rag->fe_phase, sizeof(p), hipMemcpyDeviceToHost));   // the same for every channel of the handle
  *pending = p & 7u;
  return HRFD_OK;
}

// ------------------------------------------------------------------ inner boundary
// hrfd_demod: n_channels instances of ONE demodulator class, fed with the
// 256 kS/s, already mixed, int8 IQ stream -- X::acceptIqData(int8_t*,uint32_t).
// Same kernels as the outer boundary, entered behind the front end (src256).
struct hrfd_demod
{
  hrfd_rx *rx = nullptr;
  int mode = 0;
};

extern "C" int hrfd_demod_create(int mode, uint32_t n_channels, int device, hrfd_demod **out)
{
  if (out == nullptr || mode < HRFD_MODE_AM || mode > HRFD_MODE_USB)
  {
    return fail(HRFD_EINVAL, "hrfd_demod_create: mode must be AM, FM, WBFM, LSB or USB");
  }
  *out = nullptr;
  hrfd_rx *rx = nullptr;
  int rc = hrfd_rx_create(n_channels, device, &rx);
  if (rc != HRFD_OK)
  {
    return rc;
  }
  rc = hrfd_rx_set_mode(rx, HRFD_ALL_CHANNELS, mode);
  if (rc != HRFD_OK)
  {
    rx_free(rx);
    return rc;
  }
  hrfd_demod *h = new hrfd_demod;
  h->rx = rx;
  h->mode = mode;
  *out = h;
  return HRFD_OK;
}

extern "C" int hrfd_demod_destroy(hrfd_demod *h)
{
  if (h != nullptr)
  {
    rx_free(h->rx);
    delete h;
  }
  return HRFD_OK;
}

extern "C" int hrfd_demod_reset(hrfd_demod *h, uint32_t channel)
{
  if (h == nullptr)
  {
    return fail(HRFD_EINVAL, "NULL handle");
  }
  return hrfd_rx_reset_demod(h->rx, channel, h->mode);
}

extern "C" int hrfd_demod_set_gain(hrfd_demod *h, uint32_t channel, float gain)
{
  if (h == nullptr)
  {
    return fail(HRFD_EINVAL, "NULL handle");
  }
  return hrfd_rx_set_gain(h->rx, channel, h->mode, gain);
}

extern "C" int hrfd_demod_set_sideband(hrfd_demod *h, uint32_t channel, int lsb)
{
  if (h == nullptr)
  {
    return fail(HRFD_EINVAL, "NULL handle");
  }
  if (h->mode != HRFD_MODE_LSB && h->mode != HRFD_MODE_USB)
  {
    return fail(HRFD_ESTATE, "hrfd_demod_set_sideband: not an SSB demodulator");
  }
  // SsbDemodulator::set{Lsb,Usb}DemodulationMode (SsbDemodulator.cc): a flag, no state change
  return hrfd_rx_set_mode(h->rx, channel, lsb ? HRFD_MODE_LSB : HRFD_MODE_USB);
}

extern "C" int hrfd_demod_process(hrfd_demod *dh, const int8_t *iq256k, uint32_t bytes_per_channel,
                                  int16_t *pcm, uint32_t *n_pcm)
{
  if (dh == nullptr || iq256k == nullptr || pcm == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_demod_process: NULL argument");
  }
  hrfd_rx *h = dh->rx;
  if (bytes_per_channel == 0 || (bytes_per_channel & 1u) != 0 || bytes_per_channel > 32768u)
  {
    return fail(HRFD_EINVAL, "hrfd_demod_process: bytes_per_channel must be even, > 0 and <= 32768 (got %u)", bytes_per_channel);
  }
  HIP_TRY(hipSetDevice(h->device));
  const uint32_t C = h->n_channels;
  const uint32_t npcm = (bytes_per_channel + 63u) / 64u;   // hrfd_demod_pcm_capacity
  const size_t iq_bytes = (size_t)C * bytes_per_channel;
  const size_t pcm_bytes = (size_t)C * npcm * sizeof(int16_t);
  hipStream_t s = h->stream;
  int rc;
  HIP_TRY(hipStreamSynchronize(s));
  if ((rc = grow((void **)&h->d_iq, &h->cap_iq, iq_bytes)) != HRFD_OK) return rc;
  if ((rc = grow((void **)&h->d_pcm, &h->cap_pcm, pcm_bytes)) != HRFD_OK) return rc;
  if ((rc = grow((void **)&h->d_npcm, &h->cap_npcm, (size_t)C * 4)) != HRFD_OK) return rc;
  HIP_TRY(hipMemsetAsync(h->d_pcm, 0, pcm_bytes, s));
  HIP_TRY(hipMemcpyAsync(h->d_iq, iq256k, iq_bytes, hipMemcpyHostToDevice, s));
  const LaunchOpts opt = {1, 0, 0, 1};
  rc = rx_launch(h, h->d_iq, bytes_per_channel, bytes_per_channel, 1, 0, h->d_pcm, h->d_npcm, nullptr,
                 nullptr, nullptr, s, opt);
  if (rc != HRFD_OK) return rc;
  uint32_t viol = 0;
  if ((rc = hrfd_rx_sync(h, &viol)) != HRFD_OK) return rc;
  if (viol != 0)
  {
    return fail(HRFD_ESTATE, "internal: single-block launch reported %u violations", viol);
  }
  HIP_TRY(hipMemcpyAsync(pcm, h->d_pcm, pcm_bytes, hipMemcpyDeviceToHost, s));
  if (n_pcm != nullptr)
  {
    HIP_TRY(hipMemcpyAsync(n_pcm, h->d_npcm, (size_t)C * 4, hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  return HRFD_OK;
}
