// hackrfdiags_amd/csrc/hrfd_api_debug.hip -- the hrfd_*_debug_* entry points (include/hrfd_debug.h): read-only
// introspection and measurement, and the behaviour-changing test hooks that are inert without HRFD_DEBUG_HOOKS=1.
// Part of the unity translation unit hrfd_lib.hip, behind hrfd_api.hip and hrfd_api_tx.hip (the handles' structs).
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
// host only (no device needed): the first-quadrant table as hrfd_rx_create builds and proves it (build_atan_quadrant):
// out[16644] words, *ok = the proof's verdict on this host's libm
extern "C" int hrfd_debug_atan2_quadrant(uint32_t *out16644, int *ok)
{
  if (out16644 == nullptr || ok == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_debug_atan2_quadrant: NULL");
  }
  std::vector<float> lut(65536);
  build_atan2(lut.data());
  *ok = build_atan_quadrant(lut.data(), out16644) ? 1 : 0;
  return HRFD_OK;
}
// ... and the first-quadrant table of the re-split flow kernel (theta_quad)
extern "C" int hrfd_rx_debug_atan_eval_quad(hrfd_rx *h, float *out65536)
{
  if (h == nullptr || out65536 == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_rx_debug_atan_eval_quad: NULL");
  }
  if (!h->quad_ok)
  {
    return fail(HRFD_ESTATE, "hrfd_rx_debug_atan_eval_quad: the first-quadrant table did not verify against the reference table on this host");
  }
  HIP_TRY(hipSetDevice(h->device));
  float *d = nullptr;
  HIP_TRY(hipMalloc((void **)&d, sizeof(float) * 65536));
  hipLaunchKernelGGL(k_atan_eval_quad, dim3(256), dim3(256), 0, 0, h->d_atquad, d);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpy(out65536, d, sizeof(float) * 65536, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess)
  {
    return fail(HRFD_ENODEV, "hrfd_rx_debug_atan_eval_quad: %s", hipGetErrorString(e));
  }
  return HRFD_OK;
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
// four-generation rings, ..., 15 a block's magnitude slot: block b - 16 finished) as expired the first
// time it polls it -- the bounded-spin failure path (kFailExpired, abort word, host replay of the channel) on demand
// (where = 1000 p + g: no wait expires; the service wave of generation g of workgroup 0 is held up behind hand-over point p
// of its loop instead: flow_hold_up)
extern "C" int hrfd_rx_debug_expire(hrfd_rx *h, int where)
{
  HRFD_HOOK_GATE("hrfd_rx_debug_expire");
  // (12 .. 14 belong to build flags that are off; 15 is the shipped build's wait of a block b >= 16 for block b - 16's
  //  magnitude slot, hrfd_rx_flow.hip kCtlBlk: ADVICE round 5)
  if (h == nullptr || where < 0 || (where > 15 && where < 1000) || where > 10063)
  {
    return fail(HRFD_EINVAL, "hrfd_rx_debug_expire: 0 (off) .. 15, or 1000 point + generation");
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

extern "C" int hrfd_rx_debug_ragged(hrfd_rx *h, int *offgrid, unsigned long long *launches)
{
  if (h == nullptr || offgrid == nullptr || launches == nullptr)
  {
    return fail(HRFD_EINVAL, "NULL");
  }
  *offgrid = h->offgrid ? 1 : 0;
  *launches = h->ragged_launches;
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

// test hook: 1 = the recurrence on k_phase_scan<64> / k_phase_scan_plain whatever the bank size (0: k_phase_rows up to 8192 channels)
extern "C" int hrfd_mod_debug_set_scan(hrfd_mod *h, int kind)
{
  HRFD_HOOK_GATE("hrfd_mod_debug_set_scan");
  if (h == nullptr || kind < 0 || kind > 2)
  {
    return fail(HRFD_EINVAL, "hrfd_mod_debug_set_scan: kind 0 | 1 | 2");
  }
  h->scan_kind = kind;                                      // (2, round 6: k_phase_rows -- four steps per lane -- where k_phase_rows8 would run)
  return HRFD_OK;
}

// test hook: 0 = the WBFM modulator's lookup pass and x8 cascade as two kernels (k_wb_rails, k_mod<WB_TAIL>: rounds 2-5), 1 = k_wb_tail
extern "C" int hrfd_mod_debug_set_tail(hrfd_mod *h, int kind)
{
  HRFD_HOOK_GATE("hrfd_mod_debug_set_tail");
  if (h == nullptr || (kind != 0 && kind != 1))
  {
    return fail(HRFD_EINVAL, "hrfd_mod_debug_set_tail: kind 0 | 1");
  }
  h->wb_fused = kind;
  return HRFD_OK;
}
