// hackrfdiags_amd/csrc/hrfd_api_tx.hip -- the transmit side of the C ABI (include/hrfd.h): hrfd_mod_* (the four modulators,
// interpolateSignal and the signals/ generators over k_mod and its baseband passes) and hrfd_nco_*.  Part of the unity
// translation unit hrfd_lib.hip, behind hrfd_api.hip (errors, HIP_TRY, grow).
// ------------------------------------------------------------------ transmit
#ifndef HRFD_WB_FUSED
#define HRFD_WB_FUSED 1                 // round 6: the WBFM modulator's lookup pass and x8 cascade as ONE kernel (k_wb_tail)
#endif
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
  int scan_kind = 0;                     // test hook: 1 = k_phase_scan<64> / k_phase_scan_plain whatever the bank size, 2 = k_phase_rows (four steps per lane) where k_phase_rows8 would run
  int wb_fused = HRFD_WB_FUSED;          // test hook: 0 = the lookup pass and the x8 cascade as two kernels (rounds 2-5), 1 = k_wb_tail
  // staging for the host entry
  int16_t *d_in = nullptr;
  int8_t *d_out = nullptr;
  size_t cap_in = 0, cap_out = 0;
};

// the WBFM modulator's last pass over the slice [lo, lo + len) of every channel (len 0: the whole call): Nco::runFast's
// lookup and stages 6-8 of the cascade in ONE kernel (k_wb_tail, round 6) -- or round 5's two, k_wb_rails in place over the
// phases and k_mod<WB_TAIL> (h->wb_fused == 0: the A/B and the tests that run both)
static void wb_tail_launch(hrfd_mod *h, const BaseParams &B0, const ModParams &T0, uint32_t lo, uint32_t len, uint32_t max_wgs, hipStream_t s)
{
  const uint32_t n = B0.n, C = B0.n_channels;
  const uint32_t span = (len != 0u) ? len : n;
  if (h->wb_fused != 0)
  {
    WbTailParams P;
    P.cells = h->d_wb;
    P.out = T0.out;
    P.wbpack = h->d_wbpack;
    P.wbtail = T0.wbtail;
    P.wbtail_out = B0.wbtail_out;
    P.n = n;
    P.n_channels = C;
    P.lo = lo;
    P.len = len;
    const uint32_t runs = (span * 32u + kWtRun - 1u) / kWtRun;
    const uint32_t items_x = ((C + 7u) / 8u) * runs;                 // work items of the fullest XCD
    const uint32_t waves = kWtThreads / 64;
    const uint32_t wgs_x = std::max(1u, std::min(max_wgs / 8u, (items_x + waves - 1u) / waves));
    hipLaunchKernelGGL(k_wb_tail, dim3(8u * wgs_x), dim3(kWtThreads), 0, s, P);
    return;
  }
  BaseParams B = B0;
  B.lo = lo;
  B.len = len;
  const size_t q = (size_t)span * 32 / 4 * C;
  hipLaunchKernelGGL(k_wb_rails, dim3((uint32_t)std::min<size_t>(max_wgs, (q + kWbRailsThreads - 1) / kWbRailsThreads)), dim3(kWbRailsThreads), 0, s, B);   // (two workgroups per CU: the 64 KiB table)
  ModParams T = T0;
  const uint32_t groups8 = 8u * ((C + 7u) / 8u);
  if (len != 0u)
  {
    T.tile0 = lo / kModTile;
    T.tiles_launch = (len + kModTile - 1) / kModTile;
  }
  const uint32_t tl = (T.tiles_launch != 0u) ? T.tiles_launch : (n + kModTile - 1) / kModTile;
  hipLaunchKernelGGL(k_mod<HRFD_MOD_WB_TAIL>, dim3(groups8 * tl), dim3(kModThreads), 0, s, T);
}

#ifndef HRFD_PHASE_ROWS8
#define HRFD_PHASE_ROWS8 1
#endif
// the Nco phase recurrence over `steps` cells per channel, rows `row_stride` cells apart: k_phase_rows (four channels per
// wave, a lone wave per SIMD up to 4096 channels, two up to 8192: the wave's time per step is the same; whole chunks of 64
// steps, which every call and every time slice is); banks beyond that put more waves on a SIMD than that shape likes and
// keep round 2's k_phase_scan<64> (64 channels per recurrence wave)
static void phase_scan(hrfd_mod *h, uint32_t *cells, size_t steps, size_t row_stride, float *d_acc, uint32_t n_channels, hipStream_t s)
{
  if (h->scan_kind != 1 && n_channels <= 8192u && (steps & 63) == 0 && row_stride < ((size_t)1 << 28))
  {
    // (round 6: eight steps per lane where the step count allows whole chunks of 128 -- every WBFM call and time slice;
    //  scan_kind 2, a test hook, keeps the four-step kernel)
    if (h->scan_kind != 2 && HRFD_PHASE_ROWS8 && (steps & 127) == 0)
    {
      hipLaunchKernelGGL(k_phase_rows8, dim3((n_channels + 15) / 16), dim3(kPrThreads), 0, s, cells, steps, row_stride, d_acc, n_channels);
    }
    else
    {
      hipLaunchKernelGGL(k_phase_rows, dim3((n_channels + 15) / 16), dim3(kPrThreads), 0, s, cells, steps, row_stride, d_acc, n_channels);
    }
  }
  else if ((steps & 3) == 0 && (row_stride & 3) == 0)
  {
    hipLaunchKernelGGL(k_phase_scan<64>, dim3((n_channels + 63) / 64), dim3(kPsThreads), 0, s, cells, steps, row_stride, d_acc, n_channels, h->d_err);
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

// Which build of glibc's sinf / cosf does this host run?  x86-64 glibc dispatches between a plain and an -mfma build of
// the same source; they differ on 34 floats with |x| < 120 (all above 17: tools/proofs/sincosf_glibc.c).  1: the fused
// build (what an FMA-capable CPU gets), 0: the plain one.  The device restates that one (glibc_sinf / glibc_cosf).
// MEASURED AND SWITCHED OFF: the FM modulator's cos / sin pass (k_fm_rails) as part of the cascade's stage-0 load
// (k_mod<FM_PHASE>: VERDICT round 4, item 5).  Bit-exact, but 0.958 ms against 0.940 for the pass on the second stream
// (profiles/r5_fmmod_fused_ab_LOSES.txt, alternating runs): a tile also makes the rails of the 64 samples of history in
// front of it, so every cos / sin is evaluated twice, and in the cascade's own waves instead of beside them.
#ifndef HRFD_FM_FUSED
#define HRFD_FM_FUSED 0
#endif
// Which build of glibc's sinf / cosf the host's libm is (glibc_sincosf on the device follows it, hrfd_tx_kernels.hip):
// 1 = the -mfma build x86-64 glibc dispatches to on an FMA-capable CPU, 0 = without fused multiply-adds, -1 = neither
// (a libm that is not glibc's, another architecture's): the device then follows the FMA build and the outputs that go
// through cosf / sinf in the reference (FM modulator, Nco::run, pm / fm generators) may differ from THAT host's libm by
// +-1 LSB -- visible through hrfd_libm_variant() (bench.py prints it, the tests key their tolerance on it).  Probed once.
static int libm_probe()
{
  static const int v = [] {
    auto f = [](uint32_t u) { float x; memcpy(&x, &u, 4); return x; };
    auto b = [](float x) { uint32_t u; memcpy(&u, &x, 4); return u; };
    volatile float x1 = f(0x418a3adbu), x2 = f(0x4255b0a9u);
    const uint32_t c = b(cosf(x1)), s = b(sinf(x2));
    if (c == 0xb7b4f770u && s == 0xbc7d08a9u) return 1;
    if (c == 0xb7b4f76fu && s == 0xbc7d08a8u) return 0;
    return -1;
  }();
  return v;
}
static int libm_variant()
{
  const int v = libm_probe();
  return v < 0 ? 1 : v;
}
extern "C" int hrfd_libm_variant(void)
{
  return libm_probe();
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
    B.libm_fma = libm_variant();
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
      // (round 6: the lengths below are in UNITS of 64 samples whatever the cascade's tile -- 64 until round 5, 128 now --
      //  and every one of them is even, so a cut is a multiple of 128 samples: a whole tile, and a whole chunk of
      //  k_phase_rows8's 128 steps x 32)
      constexpr uint32_t kUnit = 64u;
      static_assert(kModTile == 64 || kModTile == 128, "the cuts below fall on multiples of 128 samples");
      const uint32_t nt = ((n_per_channel + kUnit - 1) / kUnit) & ~1u;   // (even; the last slice ends with the call whatever is left)
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
          uint32_t len = (mid + k - 1u) / k;
          len = std::min(mid, (len + 1u) & ~1u);                // (an even number of units: see above; the last piece takes what is left)
          lens.push_back(len);
          mid -= len;
        }
        lens.insert(lens.end(), tail, tail + n_tail);
      }
      uint32_t lo = 0;
      for (size_t k = 0; k + 1 < lens.size(); k++)
      {
        lo += lens[k] * kUnit;
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
      M.in = reinterpret_cast<const int16_t *>(h->d_wb);
      M.wbtail = h->d_wbtail[h->cur];
      wb_tail_launch(h, B, M, 0u, 0u, 512u, s);
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
        (void)tl;
        wb_tail_launch(h, B, T, lo, len, 384u, h->s_tail);
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
    B.libm_fma = libm_variant();
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
      const uint32_t nt = (n_per_channel + 63u) / 64u;          // (units of 64 samples, whatever the cascade's tile: round 6)
      const bool fm_sliced = h->sliced != 0 && nt >= 64u && h->s_scan != nullptr;
      if (!fm_sliced)
      {
        hipLaunchKernelGGL(k_fm_step, dim3(gs), dim3(256), 0, s, B);
        phase_scan(h, reinterpret_cast<uint32_t *>(h->d_phase), (size_t)n_per_channel, (size_t)n_per_channel, h->d_acc, h->n_channels, s);
        // (round 5: cos / sin, x 16000 and the narrowing happen in the cascade's stage-0 load -- k_mod<FM_PHASE> reads the
        //  phases; rounds 1-4 ran a pass of its own, k_fm_rails, in front)
#if HRFD_FM_FUSED
        M.in = reinterpret_cast<const int16_t *>(h->d_phase);
        M.libm_fma = libm_variant();
        hipLaunchKernelGGL(k_mod<HRFD_MOD_FM_PHASE>, dim3(grid), dim3(kModThreads), 0, s, M);
#else
        hipLaunchKernelGGL(k_fm_rails, dim3(gs), dim3(256), 0, s, B);
        M.in = h->d_rails;
        hipLaunchKernelGGL(k_mod<HRFD_MOD_RAILS>, dim3(grid), dim3(kModThreads), 0, s, M);
#endif
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
        const uint32_t l0 = (std::max(8u, nt / 16u) + 1u) & ~1u, l1 = std::min(3u * l0 + l0 / 2u, nt - l0 - 2u) & ~1u;   // (even: cuts on whole tiles of 64 or 128)
        const uint32_t cut[4] = {0u, l0 * 64u, (l0 + l1) * 64u, n_per_channel};
        hipLaunchKernelGGL(k_fm_step, dim3(gs), dim3(256), 0, s, B);
        HIP_TRY(hipEventRecord(h->ev_fork, s));
        HIP_TRY(hipStreamWaitEvent(h->s_scan, h->ev_fork, 0));
        for (int k = 0; k < 3; k++)
        {
          const uint32_t lo = cut[k], len = cut[k + 1] - lo;
          // (the slices' recurrences follow each other in stream order: the accumulators carry over in d_acc)
          phase_scan(h, reinterpret_cast<uint32_t *>(h->d_phase) + lo, (size_t)len, (size_t)n_per_channel, h->d_acc, h->n_channels, h->s_scan);
#if !HRFD_FM_FUSED
          B.lo = lo;
          B.len = len;
          hipLaunchKernelGGL(k_fm_rails, dim3((uint32_t)(((size_t)len * h->n_channels + 255) / 256)), dim3(256), 0, h->s_scan, B);
#endif
          HIP_TRY(hipEventRecord(h->ev_scan[k], h->s_scan));
        }
        // (round 5: no k_fm_rails beside the cascade any more -- the cascade's stage-0 load makes the rails from the phases)
        M.in = HRFD_FM_FUSED ? reinterpret_cast<const int16_t *>(h->d_phase) : h->d_rails;
        M.libm_fma = libm_variant();
        for (int k = 0; k < 3; k++)
        {
          HIP_TRY(hipStreamWaitEvent(s, h->ev_scan[k], 0));
          M.tile0 = cut[k] / kModTile;
          M.tiles_launch = (cut[k + 1] - cut[k] + kModTile - 1) / kModTile;
          if (HRFD_FM_FUSED) hipLaunchKernelGGL(k_mod<HRFD_MOD_FM_PHASE>, dim3(groups8 * M.tiles_launch), dim3(kModThreads), 0, s, M);
          else hipLaunchKernelGGL(k_mod<HRFD_MOD_RAILS>, dim3(groups8 * M.tiles_launch), dim3(kModThreads), 0, s, M);
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
    B.libm_fma = libm_variant();
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
  N.libm_fma = libm_variant();
  hipLaunchKernelGGL(k_nco, dim3((h->n_channels + 63) / 64), dim3(64), 0, h->stream, N);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(i_out, h->d_i, bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(q_out, h->d_q, bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return HRFD_OK;
}
