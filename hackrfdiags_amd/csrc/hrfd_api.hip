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
};

static int rx_free(hrfd_rx *h)
{
  if (h == nullptr)
  {
    return HRFD_OK;
  }
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  void *ptrs[] = {h->d_cfg, h->d_state, h->d_state_out, h->d_lut, h->d_atcorr, h->d_atinv, h->d_atcorr2, h->d_att0, h->d_dbfs, h->d_counters,
                  h->d_lists, h->d_sub_lists, h->d_chan, h->d_present, h->d_magnitude, h->d_chk_pub, h->d_chk_spec,
                  h->d_iq, h->d_pcm, h->d_iq256, h->d_npcm, h->d_allowed, h->d_mag_out, h->d_ssb_iq, h->d_dbg};
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

// ---------------------------------------------------------------------------------------------
// The hrfd_*_debug_* entry points (include/hrfd_debug.h).  Two kinds:
//   * read-only introspection and measurement (counters, kernel times, table evaluations): always available --
//     bench.py's roofline figure comes from hrfd_rx_debug_kernel_ms;
//   * hooks that CHANGE what the product does (another kernel, a shrunk warm-up, an expired wait or a held-up wave on
//     purpose, the gated pass off, unsliced modulators): the test suite's means of forcing the failure and fallback paths.
//     They are inert in a process that did not ask for them: without HRFD_DEBUG_HOOKS=1 in the environment (read once, at
//     the first call) they return HRFD_ESTATE and change nothing, so a host application cannot be flipped onto those paths
//     through the shipped library by accident or by a stray symbol lookup.
// ---------------------------------------------------------------------------------------------
static bool debug_hooks_enabled()
{
  static const bool on = [] {
    const char *e = getenv("HRFD_DEBUG_HOOKS");
    return e != nullptr && e[0] == '1' && e[1] == 0;
  }();
  return on;
}
#define HRFD_HOOK_GATE(name)                                                                                         \
  do                                                                                                                 \
  {                                                                                                                  \
    if (!debug_hooks_enabled())                                                                                      \
    {                                                                                                                \
      return fail(HRFD_ESTATE, name ": behaviour-changing test hooks are off (set HRFD_DEBUG_HOOKS=1 in the environment)"); \
    }                                                                                                                \
  } while (0)

// test hook: the arithmetic atan2 evaluated on the device for all 65536 (q, i) pairs, in the
// layout of hrfd_atan2_table(); must equal that table bit for bit when the corrections fit
static int atan_eval(hrfd_rx *h, float *out65536, bool tab);
extern "C" int hrfd_rx_debug_atan_eval(hrfd_rx *h, float *out65536)
{
  return atan_eval(h, out65536, false);
}
// ... and the first-octant-table variant (theta_tab, k_rx_wbfm_flow)
extern "C" int hrfd_rx_debug_atan_eval_tab(hrfd_rx *h, float *out65536)
{
  return atan_eval(h, out65536, true);
}
static int atan_eval(hrfd_rx *h, float *out65536, bool tab)
{
  if (h == nullptr || out65536 == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_rx_debug_atan_eval: NULL");
  }
  if (tab ? !h->tab_ok : !h->arith_ok)
  {
    return fail(HRFD_ESTATE, "hrfd_rx_debug_atan_eval: the atan2 corrections do not fit 2 bits on this device");
  }
  HIP_TRY(hipSetDevice(h->device));
  float *d = nullptr;
  HIP_TRY(hipMalloc((void **)&d, sizeof(float) * 65536));
  if (tab)
  {
    hipLaunchKernelGGL(k_atan_eval<true>, dim3(256), dim3(256), 0, 0, h->d_atcorr2, h->d_att0, d);
  }
  else
  {
    hipLaunchKernelGGL(k_atan_eval<false>, dim3(256), dim3(256), 0, 0, h->d_atcorr, h->d_atinv, d);
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpy(out65536, d, sizeof(float) * 65536, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess)
  {
    return fail(HRFD_ENODEV, "hrfd_rx_debug_atan_eval: %s", hipGetErrorString(e));
  }
  return HRFD_OK;
}

// test hook: -1 automatic (arithmetic atan2 when its corrections fit), 0 force the table gather,
// 1 require the arithmetic kernel (fails if the corrections did not fit)
extern "C" int hrfd_rx_debug_set_atan(hrfd_rx *h, int mode)
{
  HRFD_HOOK_GATE("hrfd_rx_debug_set_atan");
  if (h == nullptr || mode < -1 || mode > 1)
  {
    return fail(HRFD_EINVAL, "hrfd_rx_debug_set_atan: -1, 0 or 1");
  }
  if (mode == 1 && !h->arith_ok)
  {
    return fail(HRFD_ESTATE, "hrfd_rx_debug_set_atan: the atan2 corrections do not fit 2 bits on this device");
  }
  h->atan_mode = mode;
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

// test hook (not in the public header): shrink the de-emphasis warm-up (warm / 128 tiles, at
// most kWarmTiles) and start the lanes from y = 0 instead of their seed, so that the
// speculation-failure / repair / replay paths can be exercised.  kWarm restores the default.
extern "C" int hrfd_rx_debug_set_warm(hrfd_rx *h, int warm)
{
  HRFD_HOOK_GATE("hrfd_rx_debug_set_warm");
  if (h == nullptr || warm < 0 || warm > kWarm || (warm & 1))
  {
    return fail(HRFD_EINVAL, "hrfd_rx_debug_set_warm: even, 0..%d", kWarm);
  }
  h->warm = warm;
  return HRFD_OK;
}

// measurement hook (not in the public header): bracket the demodulator kernels
// of every launch with HIP events recorded on the launch stream.  `slots` event
// pairs are used round-robin (launch i -> slot i % slots); after a sync,
// hrfd_rx_debug_kernel_ms(h, slot, &ms) returns the elapsed time of that launch.
extern "C" int hrfd_rx_debug_enable_timing(hrfd_rx *h, int slots)
{
  if (h == nullptr || slots < 0 || slots > 4096)
  {
    return fail(HRFD_EINVAL, "hrfd_rx_debug_enable_timing: 0..4096 slots");
  }
  HIP_TRY(hipSetDevice(h->device));
  for (hipEvent_t e : h->ev)
  {
    (void)hipEventDestroy(e);
  }
  h->ev.clear();
  h->ev_launches = 0;
  h->ev_seen = 0;
  // per slot: the start and the end of the launch's kernels on its stream
  for (int i = 0; i < 2 * slots; i++)
  {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    h->ev.push_back(e);
  }
  return HRFD_OK;
}

// measurement hook: bracket only every n-th launch (n >= 1; counted from the next hrfd_rx_debug_enable_timing): the
// bracketed launches fill the slots in order, the others run back to back as they do in a host that does not measure
extern "C" int hrfd_rx_debug_timing_every(hrfd_rx *h, int n)
{
  if (h == nullptr || n < 1)
  {
    return fail(HRFD_EINVAL, "hrfd_rx_debug_timing_every: n >= 1");
  }
  h->ev_every = (uint32_t)n;
  h->ev_seen = 0;
  return HRFD_OK;
}

extern "C" int hrfd_rx_debug_kernel_ms(hrfd_rx *h, int slot, float *ms)
{
  if (h == nullptr || ms == nullptr || slot < 0 || (size_t)(2 * slot + 1) >= h->ev.size())
  {
    return fail(HRFD_EINVAL, "timing slot out of range");
  }
  HIP_TRY(hipEventElapsedTime(ms, h->ev[2 * slot], h->ev[2 * slot + 1]));
  return HRFD_OK;
}

// diagnostic hook: per-workgroup cycle stamps at the phase boundaries of k_rx_wbfm<3>
// (slots 0..5; see HRFD_STAMP in hrfd_rx_kernels.hip).  cap_groups = 0 turns it off.
extern "C" int hrfd_rx_debug_stamps(hrfd_rx *h, uint32_t cap_groups, unsigned long long *host_out)
{
  if (h == nullptr)
  {
    return fail(HRFD_EINVAL, "NULL");
  }
  HIP_TRY(hipSetDevice(h->device));
  if (host_out != nullptr && h->d_dbg != nullptr)
  {
    HIP_TRY(hipMemcpy(host_out, h->d_dbg, h->dbg_cap * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return HRFD_OK;
  }
  if (h->d_dbg)
  {
    (void)hipFree(h->d_dbg);
    h->d_dbg = nullptr;
    h->dbg_cap = 0;
  }
  if (cap_groups > 0)
  {
    HIP_TRY(hipMalloc((void **)&h->d_dbg, (size_t)cap_groups * kDbgSlots * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(h->d_dbg, 0, (size_t)cap_groups * kDbgSlots * sizeof(unsigned long long)));
    h->dbg_cap = (size_t)cap_groups * kDbgSlots;
  }
  return HRFD_OK;
}

// test hook: consecutive blocks of a channel that one k_rx_wbfm workgroup walks (0 = automatic)
extern "C" int hrfd_rx_debug_set_run_len(hrfd_rx *h, int blocks)
{
  HRFD_HOOK_GATE("hrfd_rx_debug_set_run_len");
  if (h == nullptr || blocks < 0 || blocks > 64)
  {
    return fail(HRFD_EINVAL, "hrfd_rx_debug_set_run_len: 0..64");
  }
  h->run_len = blocks;
  return HRFD_OK;
}

// test hook: 0 = WBFM batches run on k_rx_wbfm (phases in sequence, two workgroups per CU) instead of
// k_rx_wbfm_flow (one persistent workgroup per CU, a continuous stream); any other value: the default
extern "C" int hrfd_rx_debug_set_stream(hrfd_rx *h, int on)
{
  HRFD_HOOK_GATE("hrfd_rx_debug_set_stream");
  if (h == nullptr)
  {
    return fail(HRFD_EINVAL, "NULL");
  }
  h->use_stream = (on == 0) ? 0 : 2;
  return HRFD_OK;
}

// test hook: workgroup 0 of the NEXT k_rx_wbfm_flow launch treats its wait number `where` (1 ring space, 2 blocks
// finished, 3 a generation's units, 4 partial sums, 5 verification order, 6 integer-stage order, 7 AM / SSB: room in the
// four-generation rings) as expired the first
// time it polls it -- the bounded-spin failure path (kFailExpired, abort word, host replay of the channel) on demand
// (where = 1000 p + g: no wait expires; the service wave of generation g of workgroup 0 is held up behind hand-over point p
// of its loop instead: flow_hold_up)
extern "C" int hrfd_rx_debug_expire(hrfd_rx *h, int where)
{
  HRFD_HOOK_GATE("hrfd_rx_debug_expire");
  if (h == nullptr || where < 0 || (where > 7 && where < 1000) || where > 8063)
  {
    return fail(HRFD_EINVAL, "hrfd_rx_debug_expire: 0 (off) .. 6");
  }
  h->expire_once = where;
  return HRFD_OK;
}

// test hook: AM / SSB / FM batches on the flow kernel's FIR modes: -1 automatic (banks of 48 channels or more per kind), 0 never, 1 always
extern "C" int hrfd_rx_debug_set_fir_flow(hrfd_rx *h, int mode)
{
  HRFD_HOOK_GATE("hrfd_rx_debug_set_fir_flow");
  if (h == nullptr || mode < -1 || mode > 2)
  {
    return fail(HRFD_EINVAL, "hrfd_rx_debug_set_fir_flow: -1, 0, 1 or 2");
  }
  h->fir_flow = mode;
  return HRFD_OK;
}

// test hook: 0 = no gated second pass on the device; a channel with a closed gate in a batch stays failed (the host replays it)
extern "C" int hrfd_rx_debug_set_gated(hrfd_rx *h, int on)
{
  HRFD_HOOK_GATE("hrfd_rx_debug_set_gated");
  if (h == nullptr)
  {
    return fail(HRFD_EINVAL, "NULL");
  }
  h->gated_pass = on ? 1 : 0;
  return HRFD_OK;
}

extern "C" int hrfd_rx_debug_set_stagger(hrfd_rx *h, int units)
{
  HRFD_HOOK_GATE("hrfd_rx_debug_set_stagger");
  if (h == nullptr || units < 0)
  {
    return fail(HRFD_EINVAL, "hrfd_rx_debug_set_stagger: 0..64");
  }
  h->stagger = units;
  return HRFD_OK;
}

// diagnostic hook: the cross-block check values of the latest launch ([n_channels][n_blocks] each)
extern "C" int hrfd_rx_debug_chk(hrfd_rx *h, float *pub, float *spec, uint32_t n)
{
  if (h == nullptr || pub == nullptr || spec == nullptr || n > h->cap_units)
  {
    return fail(HRFD_EINVAL, "hrfd_rx_debug_chk: bad arguments");
  }
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMemcpy(pub, h->d_chk_pub, n * sizeof(float), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(spec, h->d_chk_spec, n * sizeof(float), hipMemcpyDeviceToHost));
  return HRFD_OK;
}

extern "C" int hrfd_rx_debug_counters(hrfd_rx *h, uint32_t *out8)
{
  if (h == nullptr || out8 == nullptr)
  {
    return fail(HRFD_EINVAL, "NULL");
  }
  memcpy(out8, h->last_counters, sizeof(h->last_counters));
  out8[kNumCounters - 1] = h->replays;
  return HRFD_OK;
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
  if (opt.src256)
  {
    if (block_bytes == 0 || (block_bytes % 128u) != 0 || block_bytes > 32768u)
    {
      return fail(HRFD_EINVAL, "256 kS/s input must be a multiple of 128 bytes and <= 32768 (got %u)", block_bytes);
    }
  }
  else if (block_bytes == 0 || (block_bytes % 1024u) != 0 || block_bytes > HRFD_BLOCK_BYTES)
  {
    return fail(HRFD_EINVAL, "block_bytes must be a multiple of 1024 and <= %u (got %u)",
                HRFD_BLOCK_BYTES, block_bytes);
  }
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
  if (ntiles > kMaxTiles)
  {
    return fail(HRFD_EINVAL, "internal: %d de-emphasis tiles exceed %d", ntiles, kMaxTiles);
  }
  if (hal > kMaxHal)
  {
    return fail(HRFD_EINVAL, "internal: history %d exceeds %d", hal, kMaxHal);
  }
  if (n_blocks > 1 && (uint32_t)(hal + 64) * halo_unit > block_bytes)
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
  const bool flow_shape = batch && h->use_stream == 2 && h->tab_ok && h->atan_mode != 0 && (n256 % 512u) == 0 && n256 >= 2048u;
  const bool flow = flow_shape && n_wb != 0;              // the WBFM channels run on the flow kernel
  const bool fir_shape = flow_shape && h->fir_flow != 0 && n_blocks <= 64u;
  const int kinds = (n_wb != 0) + (n_as != 0) + (n_fm != 0);
  const bool bank = fir_shape && kinds >= 2 && n_blocks <= 16u && h->fir_flow != 2 && (h->fir_flow > 0 || list_count[9] >= 48u);
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
      // as long as possible (16) while the launch still fills the chip
      run_len = (h->run_len > 0) ? (uint32_t)h->run_len : 16u;
      run_len = std::min(run_len, n_blocks);
      while (h->run_len <= 0 && run_len > 1 && groups * ((n_blocks + run_len - 1) / run_len) < 256u)
      {
        run_len--;
      }
    }
    P.run_len = run_len;
    P.n_runs = (n_blocks + run_len - 1) / run_len;
    const uint32_t grid = groups * P.n_runs;
    P.dbg = (h->d_dbg != nullptr && (size_t)grid * kDbgSlots <= h->dbg_cap && mode >= 0) ? h->d_dbg : nullptr;   // (probe builds: any one mode)
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
static int rx_replay(hrfd_rx *h, const std::vector<uint32_t> &subset, const int8_t *d_iq, uint64_t stride,
                     uint32_t block_bytes, uint32_t n_blocks, uint32_t gain_db, int16_t *d_pcm, uint32_t *d_npcm,
                     uint32_t *d_mag, uint8_t *d_allowed, int8_t *d_iq256, hipStream_t s, bool pcm_is_clear)
{
  if (subset.empty())
  {
    return HRFD_OK;
  }
  // squelched units write no PCM: they must read as zeros, not as what a failed batch left there
  const size_t row = (size_t)n_blocks * (block_bytes / 512) * sizeof(int16_t);
  const bool whole_bank = subset.size() == h->n_channels;
  if (pcm_is_clear)
  {
    // (no batch ran over this buffer: the caller's memset still stands)
  }
  else if (whole_bank)
  {
    HIP_TRY(hipMemsetAsync(d_pcm, 0, row * h->n_channels, s));
  }
  else
  {
    for (uint32_t c : subset)
    {
      HIP_TRY(hipMemsetAsync(reinterpret_cast<char *>(d_pcm) + row * c, 0, row, s));
    }
  }
  for (uint32_t b = 0; b < n_blocks; b++)
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
  if (block_bytes == 0 || (block_bytes % 1024u) != 0 || block_bytes > HRFD_BLOCK_BYTES || n_blocks == 0)
  {
    return fail(HRFD_EINVAL, "block_bytes must be a multiple of 1024 and <= %u, n_blocks > 0", HRFD_BLOCK_BYTES);
  }
  HIP_TRY(hipSetDevice(h->device));
  const uint32_t C = h->n_channels;
  const size_t units = (size_t)C * n_blocks;
  const uint32_t npcm = block_bytes / 512;
  const size_t iq_bytes = units * block_bytes;
  const size_t pcm_bytes = units * npcm * sizeof(int16_t);
  const size_t iq256_bytes = units * (block_bytes / 8);
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
  std::vector<uint32_t> redo;                              // channels to run on the exact per-block path
  if (n_blocks > 1 && (uint32_t)(kMaxHal + 64) * 16u <= block_bytes)
  {
    // whole batch in one launch, blocks of a channel in parallel (speculative)
    const LaunchOpts opt = {n_blocks, 0, 0, 0};
    rc = rx_launch(h, h->d_iq, stride, block_bytes, n_blocks, gain_db, h->d_pcm, h->d_npcm,
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
  const bool batch_ran = (n_blocks > 1 && (uint32_t)(kMaxHal + 64) * 16u <= block_bytes);
  if ((rc = rx_replay(h, redo, h->d_iq, stride, block_bytes, n_blocks, gain_db, h->d_pcm, h->d_npcm, h->d_mag_out,
                      h->d_allowed, d_iq256, s, !batch_ran)) != HRFD_OK)
  {
    return rc;
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
  std::vector<int16_t> pcm((size_t)C * (block_bytes / 512 + 1));
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
  if (bytes_per_channel == 0 || (bytes_per_channel % 128u) != 0 || bytes_per_channel > 32768u)
  {
    return fail(HRFD_EINVAL, "hrfd_demod_process: bytes_per_channel must be a multiple of 128 and <= 32768");
  }
  HIP_TRY(hipSetDevice(h->device));
  const uint32_t C = h->n_channels;
  const uint32_t npcm = bytes_per_channel / 64;
  const size_t iq_bytes = (size_t)C * bytes_per_channel;
  const size_t pcm_bytes = (size_t)C * npcm * sizeof(int16_t);
  hipStream_t s = h->stream;
  int rc;
  HIP_TRY(hipStreamSynchronize(s));
  if ((rc = grow((void **)&h->d_iq, &h->cap_iq, iq_bytes)) != HRFD_OK) return rc;
  if ((rc = grow((void **)&h->d_pcm, &h->cap_pcm, pcm_bytes)) != HRFD_OK) return rc;
  if ((rc = grow((void **)&h->d_npcm, &h->cap_npcm, (size_t)C * 4)) != HRFD_OK) return rc;
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

// ------------------------------------------------------------------ transmit
struct hrfd_mod
{
  int device = 0;
  int kind = 0;
  uint32_t n_channels = 0;
  hipStream_t stream = nullptr;
  hipStream_t last_stream = nullptr;
  int16_t *d_tail[2] = {nullptr, nullptr};   // ping-pong: [C][4][kModTail]
  int cur = 0;
  uint8_t *d_lsb = nullptr;
  std::vector<uint8_t> h_lsb;
  bool lsb_dirty = true;
  std::mutex mu;
  std::vector<uint32_t> resets;
  // AM / FM: per-channel parameter (modulation index / deviation), FM phase accumulators, and
  // the baseband rails of a call
  float *d_param = nullptr, *d_acc = nullptr, *d_phase = nullptr;
  int16_t *d_rails = nullptr;
  size_t cap_phase = 0, cap_rails = 0;
  std::vector<float> h_param;
  bool param_dirty = true;
  // WBFM: the PCM at 256 kS/s, the step/phase/rails cells, Nco::runFast tables, rail history
  uint32_t *d_wb = nullptr, *d_wbtail[2] = {nullptr, nullptr};
  size_t cap_wb = 0;
  float *d_sin = nullptr, *d_cos = nullptr;
  uint32_t *d_wbpack = nullptr;         // the two tables x900 as int16 rail pairs (k_wb_rails)
  uint32_t *d_err = nullptr;            // k_phase_scan: waits that expired (never, unless the kernel is broken)
  // WBFM: the call's passes run in time slices on three streams (hrfd_mod_process_device)
  static constexpr int kMaxSlices = 32;
  hipStream_t s_scan = nullptr, s_tail = nullptr;  // the recurrence's stream; the stream of every other pass of a sliced call
  bool cu_masked = false;               // the recurrence's stream has CUs of its own
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_head[kMaxSlices] = {}, ev_scan[kMaxSlices] = {};
  int sliced = 1;                        // test hook: 0 = one pass after the other on the caller's stream
  // staging for the host entry
  int16_t *d_in = nullptr;
  int8_t *d_out = nullptr;
  size_t cap_in = 0, cap_out = 0;
};

// the Nco phase recurrence over `steps` cells per channel, rows `row_stride` cells apart (k_phase_scan: 16-byte pieces)
static void phase_scan(hrfd_mod *h, uint32_t *cells, size_t steps, size_t row_stride, float *d_acc, uint32_t n_channels, hipStream_t s)
{
  if ((steps & 3) == 0 && (row_stride & 3) == 0)
  {
    // channels per workgroup: as few as still fit the chip in one round (one workgroup per CU)
    if (n_channels <= 16u * 256u)
    {
      hipLaunchKernelGGL(k_phase_scan<16>, dim3((n_channels + 15) / 16), dim3(kPsThreads), 0, s, cells, steps, row_stride, d_acc, n_channels, h->d_err);
    }
    else if (n_channels <= 32u * 256u)
    {
      hipLaunchKernelGGL(k_phase_scan<32>, dim3((n_channels + 31) / 32), dim3(kPsThreads), 0, s, cells, steps, row_stride, d_acc, n_channels, h->d_err);
    }
    else
    {
      hipLaunchKernelGGL(k_phase_scan<64>, dim3((n_channels + 63) / 64), dim3(kPsThreads), 0, s, cells, steps, row_stride, d_acc, n_channels, h->d_err);
    }
  }
  else
  {
    hipLaunchKernelGGL(k_phase_scan_plain, dim3((n_channels + 63) / 64), dim3(64), 0, s, cells, steps, row_stride, d_acc, n_channels);
  }
}

static int mod_free(hrfd_mod *h)
{
  if (h == nullptr)
  {
    return HRFD_OK;
  }
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (hipStream_t st : {h->s_scan, h->s_tail})
  {
    if (st)
    {
      (void)hipStreamSynchronize(st);
      (void)hipStreamDestroy(st);
    }
  }
  for (hipEvent_t e : {h->ev_fork, h->ev_join})
  {
    if (e) (void)hipEventDestroy(e);
  }
  for (int i = 0; i < hrfd_mod::kMaxSlices; i++)
  {
    if (h->ev_head[i]) (void)hipEventDestroy(h->ev_head[i]);
    if (h->ev_scan[i]) (void)hipEventDestroy(h->ev_scan[i]);
  }
  void *ptrs[] = {h->d_tail[0], h->d_tail[1], h->d_lsb, h->d_in, h->d_out, h->d_param, h->d_acc, h->d_phase, h->d_rails,
                  h->d_wb, h->d_wbtail[0], h->d_wbtail[1], h->d_sin, h->d_cos, h->d_err, h->d_wbpack};
  for (void *p : ptrs)
  {
    if (p) (void)hipFree(p);
  }
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return HRFD_OK;
}

extern "C" int hrfd_mod_create(int kind, uint32_t n_channels, int device, hrfd_mod **out)
{
  if (out == nullptr || n_channels == 0 ||
      kind < HRFD_MOD_SSB || kind > HRFD_MOD_SIG_FM)
  {
    return fail(HRFD_EINVAL, "hrfd_mod_create: kind must be HRFD_MOD_SSB, _INTERP, _AM, _FM, _WBFM or _SIG_*, n_channels > 0");
  }
  *out = nullptr;
  if (hrfd_device_count() <= 0)
  {
    return fail(HRFD_ENODEV, "hrfd_mod_create: no HIP device visible (this library has no CPU path)");
  }
  if (device < 0)
  {
    HIP_TRY(hipGetDevice(&device));
  }
  HIP_TRY(hipSetDevice(device));
  hrfd_mod *h = new hrfd_mod;
  h->device = device;
  h->kind = kind;
  h->n_channels = n_channels;
  h->h_lsb.assign(n_channels, 1);                        // SsbModulator starts in LSB (SsbModulator.cc ctor)
  const size_t tail_bytes = (size_t)n_channels * 4 * kModTail * sizeof(int16_t);
  hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc((void **)&h->d_tail[0], tail_bytes);
  if (e == hipSuccess) e = hipMalloc((void **)&h->d_tail[1], tail_bytes);
  if (e == hipSuccess) e = hipMalloc((void **)&h->d_lsb, n_channels);
  if (e == hipSuccess) e = hipMemset(h->d_tail[0], 0, tail_bytes);   // zero pipelines == resetModulator()
  if (e == hipSuccess) e = hipMemset(h->d_tail[1], 0, tail_bytes);
  // AmModulator.cc:218 modulationIndex = 0.8; FmModulator.cc:218 frequencyDeviation = 3500, Nco phase 0
  // WbFmModulator.cc:204 frequencyDeviation = 70000
  h->h_param.assign(n_channels, kind == HRFD_MOD_FM ? 3500.0f : kind == HRFD_MOD_WBFM ? 70000.0f : (float)0.8);
  if (e == hipSuccess) e = hipMalloc((void **)&h->d_param, sizeof(float) * n_channels);
  if (e == hipSuccess) e = hipMalloc((void **)&h->d_acc, sizeof(float) * n_channels);
  if (e == hipSuccess) e = hipMemset(h->d_acc, 0, sizeof(float) * n_channels);
  // [0] waits that expired; [1], [2] counters of the -DHRFD_PS_PROBE diagnostic build of k_phase_scan
  if (e == hipSuccess) e = hipMalloc((void **)&h->d_err, 3 * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMemset(h->d_err, 0, 3 * sizeof(uint32_t));
  if (kind == HRFD_MOD_WBFM)
  {
    // Nco.cc:50-61: tables from a float angle accumulated by float increments; sinf/cosf: host libm
    std::vector<float> st(16384), ct(16384);
    const float inc = (float)(2 * M_PI / 16384);
    float ang = (float)(-M_PI);
    for (int i = 0; i < 16384; i++)
    {
      st[i] = sinf(ang);
      ct[i] = cosf(ang);
      ang += inc;
    }
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_sin, sizeof(float) * 16384);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_cos, sizeof(float) * 16384);
    if (e == hipSuccess) e = hipMemcpy(h->d_sin, st.data(), sizeof(float) * 16384, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->d_cos, ct.data(), sizeof(float) * 16384, hipMemcpyHostToDevice);
    // WbFmModulator.cc:604-612: iv = cos * 900 (float), (int16_t) -- per table entry instead of per sample
    std::vector<uint32_t> pack(16384);
    for (int i = 0; i < 16384; i++)
    {
      volatile float iv = ct[i] * 900.0f, qv = st[i] * 900.0f;
      pack[i] = ((uint32_t)(int)(short)(int)iv & 0xffffu) | ((uint32_t)(int)(short)(int)qv << 16);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_wbpack, sizeof(uint32_t) * 16384);
    if (e == hipSuccess) e = hipMemcpy(h->d_wbpack, pack.data(), sizeof(uint32_t) * 16384, hipMemcpyHostToDevice);
    for (int k = 0; k < 2; k++)
    {
      if (e == hipSuccess) e = hipMalloc((void **)&h->d_wbtail[k], sizeof(uint32_t) * 2 * n_channels);
      if (e == hipSuccess) e = hipMemset(h->d_wbtail[k], 0, sizeof(uint32_t) * 2 * n_channels);
    }
    // The phase recurrence runs one workgroup per 16 channels, one per CU, and every step of it is latency: a
    // workgroup of another kernel on the same CU slows it (measured: 276 -> 330..500 us per slice).  When the
    // recurrence needs at most half of the chip its stream gets CUs of its own and the other streams the rest
    // (hipExtStreamCreateWithCUMask; bit i of the mask = CU i, dealt round-robin over the XCDs by the driver).
    if (e == hipSuccess)
    {
      int cus = 0;
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
      const uint32_t scan_wgs = (n_channels + 15u) / 16u;
      const uint32_t want = (scan_wgs + 7u) / 8u * 8u;
      bool masked = false;
      if (cus >= 64 && cus <= 1024 && n_channels <= 4096u && want * 2u <= (uint32_t)cus)
      {
        const uint32_t words = ((uint32_t)cus + 31u) / 32u;
        std::vector<uint32_t> scan_mask(words, 0u), rest_mask(words, 0u);
        for (uint32_t i = 0; i < (uint32_t)cus; i++)
        {
          (i < want ? scan_mask : rest_mask)[i / 32] |= 1u << (i % 32);
        }
        hipStream_t a = nullptr, b = nullptr;
        if (hipExtStreamCreateWithCUMask(&a, words, scan_mask.data()) == hipSuccess &&
            hipExtStreamCreateWithCUMask(&b, words, rest_mask.data()) == hipSuccess)
        {
          h->s_scan = a;
          h->s_tail = b;
          masked = true;
        }
        else
        {
          (void)hipGetLastError();
          if (a) (void)hipStreamDestroy(a);
          if (b) (void)hipStreamDestroy(b);
        }
      }
      if (!masked)
      {
        e = hipStreamCreateWithFlags(&h->s_scan, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->s_tail, hipStreamNonBlocking);
      }
      h->cu_masked = masked;
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming);
    for (int k = 0; k < hrfd_mod::kMaxSlices && e == hipSuccess; k++)
    {
      e = hipEventCreateWithFlags(&h->ev_head[k], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_scan[k], hipEventDisableTiming);
    }

  }
  if (kind == HRFD_MOD_FM)
  {
    // The FM modulator's 8 kS/s phase recurrence (8192 serial steps per 16-block call: ~0.14 ms whatever the bank) runs
    // slice by slice on a stream of its own BESIDE the x256 cascade of the slice in front (hrfd_mod_process_device).  That
    // stream has the device's highest priority: the recurrence is one wave per workgroup running a dependent chain, and
    // among the cascade's thousands of workgroups it is served last and takes twice its time (measured: 108 us instead of
    // 49 for a 36-tile slice, the cascade then waits for it); with priority it runs at the rate it has alone.  (A
    // priority level also has hardware queues of its own: the stream cannot end up sharing one with the caller's stream,
    // where the two would run in submission order.)
    int lo_prio = 0, hi_prio = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);
    if (e == hipSuccess && hipStreamCreateWithPriority(&h->s_scan, hipStreamNonBlocking, hi_prio) != hipSuccess)
    {
      (void)hipGetLastError();
      h->s_scan = nullptr;                                 // (no second stream: the call runs unsliced)
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
    for (int k = 0; k < 4 && e == hipSuccess; k++)
    {
      e = hipEventCreateWithFlags(&h->ev_scan[k], hipEventDisableTiming);
    }
  }
  if (e != hipSuccess)
  {
    const int rc = fail(HRFD_ENOMEM, "hrfd_mod_create: %s", hipGetErrorString(e));
    mod_free(h);
    return rc;
  }
  *out = h;
  return HRFD_OK;
}

extern "C" int hrfd_mod_destroy(hrfd_mod *h) { return mod_free(h); }

extern "C" int hrfd_mod_reset(hrfd_mod *h, uint32_t channel)
{
  if (h == nullptr || (channel != HRFD_ALL_CHANNELS && channel >= h->n_channels))
  {
    return fail(HRFD_EINVAL, "hrfd_mod_reset: bad handle or channel");
  }
  std::lock_guard<std::mutex> g(h->mu);
  h->resets.push_back(channel);
  return HRFD_OK;
}

extern "C" int hrfd_mod_set_sideband(hrfd_mod *h, uint32_t channel, int lsb)
{
  if (h == nullptr || (channel != HRFD_ALL_CHANNELS && channel >= h->n_channels))
  {
    return fail(HRFD_EINVAL, "hrfd_mod_set_sideband: bad handle or channel");
  }
  std::lock_guard<std::mutex> g(h->mu);
  for (uint32_t c = 0; c < h->n_channels; c++)
  {
    if (channel == HRFD_ALL_CHANNELS || channel == c)
    {
      h->h_lsb[c] = lsb ? 1 : 0;
    }
  }
  h->lsb_dirty = true;
  return HRFD_OK;
}

// AmModulator::setModulationIndex (AmModulator.cc:329-336): accepted when 0 <= index <= 1
extern "C" int hrfd_mod_set_modulation_index(hrfd_mod *h, uint32_t channel, float index)
{
  if (h == nullptr || h->kind != HRFD_MOD_AM || (channel != HRFD_ALL_CHANNELS && channel >= h->n_channels))
  {
    return fail(HRFD_EINVAL, "hrfd_mod_set_modulation_index: needs an AM modulator handle and a valid channel");
  }
  std::lock_guard<std::mutex> g(h->mu);
  for (uint32_t c = 0; c < h->n_channels; c++)
  {
    if ((channel == HRFD_ALL_CHANNELS || channel == c) && (index >= 0) && (index <= 1))
    {
      h->h_param[c] = index;
    }
  }
  h->param_dirty = true;
  return HRFD_OK;
}

// FmModulator::setFrequencyDeviation (FmModulator.cc:336-346).  As in the reference the range
// test looks at the CURRENT deviation, not at the new one (kept: it is the observable behaviour).
extern "C" int hrfd_mod_set_deviation(hrfd_mod *h, uint32_t channel, float deviation)
{
  if (h == nullptr || (h->kind != HRFD_MOD_FM && h->kind != HRFD_MOD_WBFM) ||
      (channel != HRFD_ALL_CHANNELS && channel >= h->n_channels))
  {
    return fail(HRFD_EINVAL, "hrfd_mod_set_deviation: needs an FM or WBFM modulator handle and a valid channel");
  }
  const float limit = (h->kind == HRFD_MOD_FM) ? 3500.0f : 112000.0f;   // WbFmModulator.cc:313
  std::lock_guard<std::mutex> g(h->mu);
  for (uint32_t c = 0; c < h->n_channels; c++)
  {
    if ((channel == HRFD_ALL_CHANNELS || channel == c) && (h->h_param[c] >= 0) && (h->h_param[c] <= limit))
    {
      h->h_param[c] = deviation;
    }
  }
  h->param_dirty = true;
  return HRFD_OK;
}

extern "C" int hrfd_mod_process_device(hrfd_mod *h, const int16_t *d_pcm, uint32_t n_per_channel,
                                       int8_t *d_iq_out, void *stream)
{
  if (h == nullptr || d_pcm == nullptr || d_iq_out == nullptr || n_per_channel == 0)
  {
    return fail(HRFD_EINVAL, "hrfd_mod_process_device: NULL argument or n_per_channel == 0");
  }
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = (stream != nullptr) ? (hipStream_t)stream : h->stream;
  {
    std::lock_guard<std::mutex> g(h->mu);
    if (h->lsb_dirty)
    {
      HIP_TRY(hipStreamSynchronize(s));
      HIP_TRY(hipMemcpy(h->d_lsb, h->h_lsb.data(), h->n_channels, hipMemcpyHostToDevice));
      h->lsb_dirty = false;
    }
    if (h->param_dirty)
    {
      HIP_TRY(hipStreamSynchronize(s));
      HIP_TRY(hipMemcpy(h->d_param, h->h_param.data(), sizeof(float) * h->n_channels, hipMemcpyHostToDevice));
      h->param_dirty = false;
    }
    for (uint32_t ch : h->resets)
    {
      // SsbModulator::resetModulator: every pipeline back to zero
      const size_t per = (size_t)4 * kModTail * sizeof(int16_t);
      if (ch == HRFD_ALL_CHANNELS)
      {
        HIP_TRY(hipMemsetAsync(h->d_tail[h->cur], 0, per * h->n_channels, s));
        if (h->kind == HRFD_MOD_WBFM) HIP_TRY(hipMemsetAsync(h->d_wbtail[h->cur], 0, 8 * (size_t)h->n_channels, s));
        if (h->kind == HRFD_MOD_SIG_FM) HIP_TRY(hipMemsetAsync(h->d_acc, 0, sizeof(float) * h->n_channels, s));   // a fresh run of the tool
      }
      else
      {
        HIP_TRY(hipMemsetAsync(h->d_tail[h->cur] + (size_t)ch * 4 * kModTail, 0, per, s));
        if (h->kind == HRFD_MOD_WBFM) HIP_TRY(hipMemsetAsync(h->d_wbtail[h->cur] + (size_t)ch * 2, 0, 8, s));
        if (h->kind == HRFD_MOD_SIG_FM) HIP_TRY(hipMemsetAsync(h->d_acc + ch, 0, sizeof(float), s));
      }
    }
    h->resets.clear();
  }
  ModParams M;
  M.in = d_pcm;
  M.out = d_iq_out;
  M.tail_in = h->d_tail[h->cur];
  M.tail_out = h->d_tail[h->cur ^ 1];
  M.lsb = h->d_lsb;
  M.wbstep = nullptr;
  M.param = nullptr;
  M.wbtail = nullptr;
  M.n = n_per_channel;
  M.n_channels = h->n_channels;
  M.tile0 = 0;
  M.tiles_launch = 0;
  const uint32_t tiles = (n_per_channel + kModTile - 1) / kModTile;
  const uint32_t groups8 = 8u * ((h->n_channels + 7u) / 8u);     // k_mod deals channels to XCDs: whole groups of eight
  const uint32_t grid = groups8 * tiles;
  if (h->kind == HRFD_MOD_WBFM)
  {
    // WbFmModulator::acceptData (WbFmModulator.cc:341-356): x32 on the PCM, the 256 kS/s Nco, x8
    const size_t samples = (size_t)n_per_channel * h->n_channels;
    const size_t s32 = samples * 32;
    int rc;
    if (s32 * 4 > h->cap_wb)
    {
      HIP_TRY(hipStreamSynchronize(s));
      if ((rc = grow((void **)&h->d_wb, &h->cap_wb, s32 * 4)) != HRFD_OK) return rc;
    }
    // (Cutting the bank into groups of channels on streams of their own buys nothing: the recurrence's time does not
    // depend on the number of channels, so every group's recurrence runs at the same time and the per-sample passes
    // still queue up in front of and behind it -- measured, 8.8 ms either way for 1024 channels.)
    BaseParams B;
    memset(&B, 0, sizeof(B));
    B.pcm = d_pcm;
    B.rails = h->d_rails;
    B.param = h->d_param;
    B.acc = h->d_acc;
    B.wb = h->d_wb;
    B.cos_t = h->d_cos;
    B.sin_t = h->d_sin;
    B.wbpack = h->d_wbpack;
    B.wbtail_out = h->d_wbtail[h->cur ^ 1];
    B.n = n_per_channel;
    B.n_channels = h->n_channels;
    // The passes run in TIME SLICES of whole blocks (512 PCM samples), on three streams: the x32 cascade with the Nco
    // steps (k_mod<WB_HEAD>) on the caller's, the phase recurrence -- serial per channel, the same 17 ns per step for
    // 64 channels as for 4096, two thirds of the call -- on one of the handle's, the table lookup and the x8 cascade
    // (k_wb_rails, k_mod<WB_TAIL>) on another: slice t's rails and tail run beside the recurrence of slice t + 1, so
    // the call costs little more than the recurrence alone.  (WbFmModulator.cc:583-637 does the three per sample.)
    // slice boundaries (input samples, multiples of the cascade's tile): a short first slice (the recurrence starts
    // behind its head pass), short last ones (what is left behind the last recurrence is one slice's rails and
    // tail), long ones between (every slice costs the recurrence a launch: ~12 us)
    std::vector<uint32_t> cuts;
    {
      // Lengths in tiles of the cascade (64 input samples).  The recurrence takes ~0.54 us per input sample, a head
      // pass ~0.1, rails and tail together ~0.25 (on the CUs the recurrence leaves them) plus ~30 us of launches:
      // slices may grow fourfold at the start (the next head pass is through before the recurrence of the slice in
      // front is) and halve at the end (a slice's rails and tail are through before the next, shorter recurrence is);
      // what stays exposed is the first slice's head pass and the last slice's rails and tail, so those two slices
      // are two tiles long.  Every slice costs the recurrence a launch (~20 us).
      static_assert(kModTile == 64, "slice lengths below are in tiles of 64 samples");
      const uint32_t nt = (n_per_channel + kModTile - 1) / kModTile;
      std::vector<uint32_t> lens;
      if (nt > 24)
      {
        const uint32_t head[2] = {2, 8};
        const uint32_t tail4[4] = {16, 8, 4, 2}, tail2[2] = {4, 2};
        const bool long_tail = nt >= 72;
        const uint32_t n_tail = long_tail ? 4u : 2u;
        const uint32_t *tail = long_tail ? tail4 : tail2;
        uint32_t mid = nt - 10u - (long_tail ? 30u : 6u);
        lens.assign(head, head + 2);
        const uint32_t room = (uint32_t)hrfd_mod::kMaxSlices - 2u - n_tail - 1u;
        const uint32_t piece = std::max(32u, (mid + room - 1u) / room);
        while (mid != 0u)
        {
          const uint32_t k = (mid + piece - 1u) / piece;        // pieces still to go: even shares
          const uint32_t len = (mid + k - 1u) / k;
          lens.push_back(len);
          mid -= len;
        }
        lens.insert(lens.end(), tail, tail + n_tail);
      }
      uint32_t lo = 0;
      for (size_t k = 0; k + 1 < lens.size(); k++)
      {
        lo += lens[k] * kModTile;
        cuts.push_back(lo);
      }
      cuts.push_back(n_per_channel);                          // (the last slice ends with the call, whole tile or not)
    }
    // (only when the recurrence has CUs of its own: beside other kernels on its CUs it loses more than the overlap gains)
    const bool sliced = h->sliced != 0 && cuts.size() > 1 && h->s_scan != nullptr && (h->cu_masked || h->sliced > 1);
    M.in = d_pcm;                                             // (k_mod<WB_HEAD> reads the PCM itself)
    M.wbstep = h->d_wb;
    M.param = h->d_param;
    if (!sliced)
    {
      hipLaunchKernelGGL(k_mod<HRFD_MOD_WB_HEAD>, dim3(grid), dim3(kModThreads), 0, s, M);
      phase_scan(h, h->d_wb, (size_t)n_per_channel * 32, (size_t)n_per_channel * 32, h->d_acc, h->n_channels, s);
      hipLaunchKernelGGL(k_wb_rails, dim3((uint32_t)std::min<size_t>(512, (s32 / 4 + kWbRailsThreads - 1) / kWbRailsThreads)), dim3(kWbRailsThreads), 0, s, B);   // (two workgroups per CU: the 64 KiB table)
      M.in = reinterpret_cast<const int16_t *>(h->d_wb);
      M.wbtail = h->d_wbtail[h->cur];
      hipLaunchKernelGGL(k_mod<HRFD_MOD_WB_TAIL>, dim3(grid), dim3(kModThreads), 0, s, M);
    }
    else
    {
      // Two streams of the handle's own: the recurrences on one (with CUs of its own), every other pass on the second.
      // The caller's stream only forks and joins (it may share its hardware queue with either: when it carried
      // kernels, everything ran in series).  The head passes of all slices go out first -- they depend on nothing but
      // the input -- and the rails and tails of the first slices queue up behind them: those have a millisecond of
      // slack, and every stream with a CU mask is a hardware queue of its own, of which a process should hold few
      // (measured: the same call takes 4.7 ms in a process with five queues and 5.2 with seven).
      hipStream_t hs = h->s_tail;
      HIP_TRY(hipEventRecord(h->ev_fork, s));
      HIP_TRY(hipStreamWaitEvent(hs, h->ev_fork, 0));
      HIP_TRY(hipStreamWaitEvent(h->s_scan, h->ev_fork, 0));
      ModParams T = M;
      T.in = reinterpret_cast<const int16_t *>(h->d_wb);
      T.wbtail = h->d_wbtail[h->cur];
      for (size_t k = 0; k < cuts.size(); k++)
      {
        const uint32_t lo = (k == 0) ? 0u : cuts[k - 1], len = cuts[k] - lo;
        M.tile0 = lo / kModTile;
        M.tiles_launch = (len + kModTile - 1) / kModTile;
        hipLaunchKernelGGL(k_mod<HRFD_MOD_WB_HEAD>, dim3(groups8 * M.tiles_launch), dim3(kModThreads), 0, hs, M);
        HIP_TRY(hipEventRecord(h->ev_head[k], hs));
      }
      for (size_t k = 0; k < cuts.size(); k++)
      {
        const uint32_t lo = (k == 0) ? 0u : cuts[k - 1], len = cuts[k] - lo;
        const uint32_t tl = (len + kModTile - 1) / kModTile;
        // (the head passes are through long before the fourth recurrence starts: it waits for the last of them, the
        // ones behind it for nothing -- every wait is a packet the queue takes microseconds over)
        if (k < 3)
        {
          HIP_TRY(hipStreamWaitEvent(h->s_scan, h->ev_head[k], 0));
        }
        else if (k == 3)
        {
          HIP_TRY(hipStreamWaitEvent(h->s_scan, h->ev_head[cuts.size() - 1], 0));
        }
        phase_scan(h, h->d_wb + (size_t)lo * 32, (size_t)len * 32, (size_t)n_per_channel * 32, h->d_acc, h->n_channels, h->s_scan);
        HIP_TRY(hipEventRecord(h->ev_scan[k], h->s_scan));
        HIP_TRY(hipStreamWaitEvent(h->s_tail, h->ev_scan[k], 0));
        B.lo = lo;
        B.len = len;
        const size_t q = (size_t)len * 32 / 4 * h->n_channels;
        hipLaunchKernelGGL(k_wb_rails, dim3((uint32_t)std::min<size_t>(384, (q + kWbRailsThreads - 1) / kWbRailsThreads)), dim3(kWbRailsThreads), 0, h->s_tail, B);
        T.tile0 = lo / kModTile;
        T.tiles_launch = tl;
        hipLaunchKernelGGL(k_mod<HRFD_MOD_WB_TAIL>, dim3(groups8 * tl), dim3(kModThreads), 0, h->s_tail, T);
      }
      HIP_TRY(hipEventRecord(h->ev_join, h->s_tail));
      HIP_TRY(hipStreamWaitEvent(s, h->ev_join, 0));
      M.tile0 = 0;
      M.tiles_launch = 0;
    }
  }
  else if (h->kind == HRFD_MOD_AM || h->kind == HRFD_MOD_FM)
  {
    // baseband rails first (k_am_rails / k_fm_phase + k_fm_rails), then the shared x256 cascade
    const size_t samples = (size_t)n_per_channel * h->n_channels;
    int rc;
    if (samples * 4 > h->cap_rails || (h->kind == HRFD_MOD_FM && samples * 4 > h->cap_phase))
    {
      HIP_TRY(hipStreamSynchronize(s));
      if ((rc = grow((void **)&h->d_rails, &h->cap_rails, samples * 4)) != HRFD_OK) return rc;
      if (h->kind == HRFD_MOD_FM && (rc = grow((void **)&h->d_phase, &h->cap_phase, samples * 4)) != HRFD_OK) return rc;
    }
    BaseParams B;
    memset(&B, 0, sizeof(B));
    B.pcm = d_pcm;
    B.rails = h->d_rails;
    B.param = h->d_param;
    B.acc = h->d_acc;
    B.phase = h->d_phase;
    B.n = n_per_channel;
    B.n_channels = h->n_channels;
    const uint32_t gs = (uint32_t)((samples + 255) / 256);
    if (h->kind == HRFD_MOD_AM)
    {
      hipLaunchKernelGGL(k_am_rails, dim3(gs), dim3(256), 0, s, B);
    }
    else
    {
      // FmModulator::modulateSignal (FmModulator.cc:586-627) sets the Nco's frequency and runs it once per PCM sample: the
      // step of every sample in parallel (k_fm_step), the phase recurrence (serial per channel: k_phase_scan), cos / sin of
      // every phase in parallel (k_fm_rails), then the cascade.  Only the recurrence is serial in time, and it is a quarter
      // of the cascade's time per sample: a long call is cut into three TIME SLICES and the recurrence and rails of slice
      // k + 1 run on a stream of their own beside the cascade of slice k.  What stays exposed is the first slice's
      // recurrence and rails.  (Round 3 ran the four passes one after the other: the recurrence's 0.14 ms and the rails'
      // 0.04 sat in front of the cascade's 0.81.)
      const uint32_t nt = tiles;
      const bool fm_sliced = h->sliced != 0 && nt >= 64u && h->s_scan != nullptr;
      if (!fm_sliced)
      {
        hipLaunchKernelGGL(k_fm_step, dim3(gs), dim3(256), 0, s, B);
        phase_scan(h, reinterpret_cast<uint32_t *>(h->d_phase), (size_t)n_per_channel, (size_t)n_per_channel, h->d_acc, h->n_channels, s);
        hipLaunchKernelGGL(k_fm_rails, dim3(gs), dim3(256), 0, s, B);
        M.in = h->d_rails;
        hipLaunchKernelGGL(k_mod<HRFD_MOD_RAILS>, dim3(grid), dim3(kModThreads), 0, s, M);
      }
      else
      {
        // Slice lengths in tiles of 64 samples: the recurrence + cos / sin of a slice take ~1.9 us per tile, the cascade
        // ~5.9 us per tile: a slice may be three times the one in front.  The recurrences and rails of ALL slices follow
        // each other on s_scan; the caller's stream carries the steps and the cascade launches, each behind its slice's
        // event.  Order matters more than priority here: a recurrence workgroup is seven waves, a cascade workgroup four,
        // and once a cascade launch has filled the CUs the slots it frees are retaken four waves at a time -- the
        // recurrence launched BEHIND a cascade launch waits for room and takes twice its time (measured: 108 us for a
        // 36-tile slice instead of 49, whatever the stream's priority).  This way slice k + 1's recurrence is resident
        // before the cascade of slice k starts (its event takes ~13 us to cross queues): timeline of a step in
        // profiles/r4_fmmod_timeline.txt.  Exposed: two event hops, the first slice's recurrence and rails.
        const uint32_t l0 = std::max(8u, nt / 16u), l1 = std::min(3u * l0 + l0 / 2u, nt - l0 - 1u);
        const uint32_t cut[4] = {0u, l0 * kModTile, (l0 + l1) * kModTile, n_per_channel};
        hipLaunchKernelGGL(k_fm_step, dim3(gs), dim3(256), 0, s, B);
        HIP_TRY(hipEventRecord(h->ev_fork, s));
        HIP_TRY(hipStreamWaitEvent(h->s_scan, h->ev_fork, 0));
        for (int k = 0; k < 3; k++)
        {
          const uint32_t lo = cut[k], len = cut[k + 1] - lo;
          // (the slices' recurrences follow each other in stream order: the accumulators carry over in d_acc)
          phase_scan(h, reinterpret_cast<uint32_t *>(h->d_phase) + lo, (size_t)len, (size_t)n_per_channel, h->d_acc, h->n_channels, h->s_scan);
          B.lo = lo;
          B.len = len;
          hipLaunchKernelGGL(k_fm_rails, dim3((uint32_t)(((size_t)len * h->n_channels + 255) / 256)), dim3(256), 0, h->s_scan, B);
          HIP_TRY(hipEventRecord(h->ev_scan[k], h->s_scan));
        }
        M.in = h->d_rails;
        for (int k = 0; k < 3; k++)
        {
          HIP_TRY(hipStreamWaitEvent(s, h->ev_scan[k], 0));
          M.tile0 = cut[k] / kModTile;
          M.tiles_launch = (cut[k + 1] - cut[k] + kModTile - 1) / kModTile;
          hipLaunchKernelGGL(k_mod<HRFD_MOD_RAILS>, dim3(groups8 * M.tiles_launch), dim3(kModThreads), 0, s, M);
        }
        M.tile0 = 0;
        M.tiles_launch = 0;
      }
    }
    if (h->kind == HRFD_MOD_AM)
    {
      M.in = h->d_rails;
      hipLaunchKernelGGL(k_mod<HRFD_MOD_RAILS>, dim3(grid), dim3(kModThreads), 0, s, M);
    }
  }
  else if (h->kind >= HRFD_MOD_SIG_AM)
  {
    // signals/{am,dsb,pm,fm}.cc | interpolateSignal: baseband pairs, then the x256 cascade with
    // interpolateSignal's own stage-1 table
    const size_t samples = (size_t)n_per_channel * h->n_channels;
    int rc;
    if (samples * 4 > h->cap_rails)
    {
      HIP_TRY(hipStreamSynchronize(s));
      if ((rc = grow((void **)&h->d_rails, &h->cap_rails, samples * 4)) != HRFD_OK) return rc;
    }
    BaseParams B;
    memset(&B, 0, sizeof(B));
    B.pcm = d_pcm;
    B.rails = h->d_rails;
    B.acc = h->d_acc;
    B.n = n_per_channel;
    B.n_channels = h->n_channels;
    const uint32_t gs = (uint32_t)((samples + 255) / 256);
    if (h->kind == HRFD_MOD_SIG_AM)
    {
      hipLaunchKernelGGL(k_sig_rails<HRFD_MOD_SIG_AM>, dim3(gs), dim3(256), 0, s, B);
    }
    else if (h->kind == HRFD_MOD_SIG_DSB)
    {
      hipLaunchKernelGGL(k_sig_rails<HRFD_MOD_SIG_DSB>, dim3(gs), dim3(256), 0, s, B);
    }
    else if (h->kind == HRFD_MOD_SIG_PM)
    {
      hipLaunchKernelGGL(k_sig_rails<HRFD_MOD_SIG_PM>, dim3(gs), dim3(256), 0, s, B);
    }
    else
    {
      hipLaunchKernelGGL(k_sig_fm, dim3((h->n_channels + 63) / 64), dim3(64), 0, s, B);
    }
    M.in = h->d_rails;
    hipLaunchKernelGGL(k_mod<HRFD_MOD_INTERP>, dim3(grid), dim3(kModThreads), 0, s, M);
  }
  else if (h->kind == HRFD_MOD_SSB)
  {
    hipLaunchKernelGGL(k_mod<HRFD_MOD_SSB>, dim3(grid), dim3(kModThreads), 0, s, M);
  }
  else
  {
    hipLaunchKernelGGL(k_mod<HRFD_MOD_INTERP>, dim3(grid), dim3(kModThreads), 0, s, M);
  }
  HIP_TRY(hipGetLastError());
  h->cur ^= 1;
  h->last_stream = s;
  return HRFD_OK;
}

// test hook: 0 = the WBFM modulator's passes one after the other on the caller's stream (no time slices)
extern "C" int hrfd_mod_debug_set_sliced(hrfd_mod *h, int on)
{
  HRFD_HOOK_GATE("hrfd_mod_debug_set_sliced");
  if (h == nullptr)
  {
    return fail(HRFD_EINVAL, "NULL");
  }
  h->sliced = on;                                          // 0 off, 1 when the recurrence's stream has CUs of its own, 2 always
  return HRFD_OK;
}

extern "C" int hrfd_mod_sync(hrfd_mod *h)
{
  if (h == nullptr)
  {
    return fail(HRFD_EINVAL, "NULL handle");
  }
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->last_stream ? h->last_stream : h->stream));
  if (h->kind == HRFD_MOD_FM || h->kind == HRFD_MOD_WBFM)
  {
    uint32_t expired = 0;
    HIP_TRY(hipMemcpy(&expired, h->d_err, sizeof(expired), hipMemcpyDeviceToHost));
    if (expired != 0)
    {
      return fail(HRFD_ESTATE, "hrfd_mod_sync: k_phase_scan gave up waiting %u time(s): the output of this handle is not valid", expired);
    }
  }
  return HRFD_OK;
}

extern "C" int hrfd_mod_process(hrfd_mod *h, const int16_t *pcm, uint32_t n_per_channel, int8_t *iq_out,
                                uint32_t *out_bytes)
{
  if (h == nullptr || pcm == nullptr || iq_out == nullptr || n_per_channel == 0)
  {
    return fail(HRFD_EINVAL, "hrfd_mod_process: NULL argument or n_per_channel == 0");
  }
  HIP_TRY(hipSetDevice(h->device));
  const size_t per_in = (size_t)n_per_channel * (h->kind == HRFD_MOD_INTERP ? 2 : 1) * sizeof(int16_t);
  const size_t in_bytes = per_in * h->n_channels;
  const size_t out_total = (size_t)h->n_channels * n_per_channel * 512;
  int rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  if ((rc = grow((void **)&h->d_in, &h->cap_in, in_bytes)) != HRFD_OK) return rc;
  if ((rc = grow((void **)&h->d_out, &h->cap_out, out_total)) != HRFD_OK) return rc;
  HIP_TRY(hipMemcpyAsync(h->d_in, pcm, in_bytes, hipMemcpyHostToDevice, h->stream));
  if ((rc = hrfd_mod_process_device(h, h->d_in, n_per_channel, h->d_out, h->stream)) != HRFD_OK) return rc;
  HIP_TRY(hipMemcpyAsync(iq_out, h->d_out, out_total, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (out_bytes != nullptr)
  {
    *out_bytes = n_per_channel << 9;                       // bytes per channel (SsbModulator.cc:512)
  }
  return HRFD_OK;
}

// ------------------------------------------------------------------ Nco
struct hrfd_nco
{
  int device = 0;
  uint32_t n_channels = 0;
  float sample_rate = 0;
  hipStream_t stream = nullptr;
  float *d_acc = nullptr, *d_step = nullptr, *d_sin = nullptr, *d_cos = nullptr;
  float *d_i = nullptr, *d_q = nullptr;
  size_t cap_out = 0;
  std::vector<float> h_step;
  bool step_dirty = true;
};

static int nco_free(hrfd_nco *h)
{
  if (h == nullptr)
  {
    return HRFD_OK;
  }
  (void)hipSetDevice(h->device);
  void *ptrs[] = {h->d_acc, h->d_step, h->d_sin, h->d_cos, h->d_i, h->d_q};
  for (void *p : ptrs)
  {
    if (p) (void)hipFree(p);
  }
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return HRFD_OK;
}

extern "C" int hrfd_nco_create(uint32_t n_channels, float sample_rate, float frequency, int device,
                               hrfd_nco **out)
{
  if (out == nullptr || n_channels == 0)
  {
    return fail(HRFD_EINVAL, "hrfd_nco_create: bad arguments");
  }
  *out = nullptr;
  if (hrfd_device_count() <= 0)
  {
    return fail(HRFD_ENODEV, "hrfd_nco_create: no HIP device visible (this library has no CPU path)");
  }
  if (device < 0)
  {
    HIP_TRY(hipGetDevice(&device));
  }
  HIP_TRY(hipSetDevice(device));
  hrfd_nco *h = new hrfd_nco;
  h->device = device;
  h->n_channels = n_channels;
  h->sample_rate = sample_rate;
  // PhaseAccumulator.cc:41: double expression stored to float
  h->h_step.assign(n_channels, (float)((2 * M_PI * frequency) / sample_rate));
  // Nco.cc:50-61: tables from a float angle accumulated by float increments; sin/cos
  // of a float argument are sinf/cosf under the C++ overloads -> host libm
  std::vector<float> st(16384), ct(16384);
  {
    const float inc = (float)(2 * M_PI / 16384);
    float ang = (float)(-M_PI);
    for (int i = 0; i < 16384; i++)
    {
      st[i] = sinf(ang);
      ct[i] = cosf(ang);
      ang += inc;
    }
  }
  hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc((void **)&h->d_acc, sizeof(float) * n_channels);
  if (e == hipSuccess) e = hipMalloc((void **)&h->d_step, sizeof(float) * n_channels);
  if (e == hipSuccess) e = hipMalloc((void **)&h->d_sin, sizeof(float) * 16384);
  if (e == hipSuccess) e = hipMalloc((void **)&h->d_cos, sizeof(float) * 16384);
  if (e == hipSuccess) e = hipMemset(h->d_acc, 0, sizeof(float) * n_channels);
  if (e == hipSuccess) e = hipMemcpy(h->d_sin, st.data(), sizeof(float) * 16384, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(h->d_cos, ct.data(), sizeof(float) * 16384, hipMemcpyHostToDevice);
  if (e != hipSuccess)
  {
    const int rc = fail(HRFD_ENOMEM, "hrfd_nco_create: %s", hipGetErrorString(e));
    nco_free(h);
    return rc;
  }
  *out = h;
  return HRFD_OK;
}

extern "C" int hrfd_nco_destroy(hrfd_nco *h) { return nco_free(h); }

extern "C" int hrfd_nco_set_frequency(hrfd_nco *h, uint32_t channel, float frequency)
{
  if (h == nullptr || (channel != HRFD_ALL_CHANNELS && channel >= h->n_channels))
  {
    return fail(HRFD_EINVAL, "hrfd_nco_set_frequency: bad handle or channel");
  }
  const float step = (float)((2 * M_PI * frequency) / h->sample_rate);   // PhaseAccumulator.cc:105
  for (uint32_t c = 0; c < h->n_channels; c++)
  {
    if (channel == HRFD_ALL_CHANNELS || channel == c)
    {
      h->h_step[c] = step;
    }
  }
  h->step_dirty = true;
  return HRFD_OK;
}

extern "C" int hrfd_nco_reset(hrfd_nco *h, uint32_t channel)
{
  if (h == nullptr || (channel != HRFD_ALL_CHANNELS && channel >= h->n_channels))
  {
    return fail(HRFD_EINVAL, "hrfd_nco_reset: bad handle or channel");
  }
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (channel == HRFD_ALL_CHANNELS)
  {
    HIP_TRY(hipMemset(h->d_acc, 0, sizeof(float) * h->n_channels));
  }
  else
  {
    HIP_TRY(hipMemset(h->d_acc + channel, 0, sizeof(float)));
  }
  return HRFD_OK;
}

extern "C" int hrfd_nco_run(hrfd_nco *h, int fast, uint32_t count, float *i_out, float *q_out)
{
  if (h == nullptr || i_out == nullptr || q_out == nullptr || count == 0)
  {
    return fail(HRFD_EINVAL, "hrfd_nco_run: bad arguments");
  }
  HIP_TRY(hipSetDevice(h->device));
  const size_t bytes = sizeof(float) * (size_t)h->n_channels * count;
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (bytes > h->cap_out)
  {
    size_t c1 = 0, c2 = 0;
    int rc;
    if ((rc = grow((void **)&h->d_i, &c1, bytes)) != HRFD_OK) return rc;
    if ((rc = grow((void **)&h->d_q, &c2, bytes)) != HRFD_OK) return rc;
    h->cap_out = bytes;
  }
  if (h->step_dirty)
  {
    HIP_TRY(hipMemcpy(h->d_step, h->h_step.data(), sizeof(float) * h->n_channels, hipMemcpyHostToDevice));
    h->step_dirty = false;
  }
  NcoParams N;
  N.acc = h->d_acc;
  N.step = h->d_step;
  N.sin_t = h->d_sin;
  N.cos_t = h->d_cos;
  N.i_out = h->d_i;
  N.q_out = h->d_q;
  N.n_channels = h->n_channels;
  N.count = count;
  N.fast = fast;
  hipLaunchKernelGGL(k_nco, dim3((h->n_channels + 63) / 64), dim3(64), 0, h->stream, N);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(i_out, h->d_i, bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(q_out, h->d_q, bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return HRFD_OK;
}
