// hackrfdiags_amd/csrc/hrfd_rx_flow.hip -- k_rx_wbfm_flow: the WBFM chain of a batch as ONE continuous
// stream per workgroup, without workgroup barriers (gfx950).
//
// Its predecessor (round 1's k_rx_wbfm_stream, removed in round 3) kept two whole blocks in LDS and met at a
// workgroup barrier once per block.  Measured on MI355X (gpurun_out/r2_ab*.log): its phases B and C were
// hidden already (removing them buys 3 %) and so is HBM; what costs is phase A running at ~60 % of
// its issue-bound rate -- every wave starts a run right behind the barrier (two dependent memory
// latencies each, all at the same time), and waves that finish early leave their SIMD to a lone
// wave, which issues at half the rate of four.  Here nothing ever meets:
//
//   waves SVC..15 ("stream")   take UNITS of two 4 KiB pieces from an LDS counter, in stream order,
//        and turn raw IQ into v (quad_piece, the phase A of hrfd_rx_kernels.hip) in a RING of
//        kFRingTiles 64-sample tiles.  The next unit's loads are issued while the current unit
//        is computed, across unit boundaries: a stream wave only ever waits when the ring is full.
//   waves 0..SVC-1 ("service") follow in GENERATIONS of 64 tiles (one per lane): geometric partial
//        sums, seed, warm-up over the two tiles in front (reads only: v is never overwritten),
//        then the tile itself, where the lane also narrows y to int16 and runs D(8,4) in registers
//        (WbFmDemodulator.cc:468-476).  Neither y nor its int16 form ever goes to LDS; a lane keeps
//        16 U samples.  Generations complete in order (b_done): verification against the left
//        neighbour's final y, repair of a tile that has not merged (re-run from the true value;
//        v is intact), first U of every tile from the LEFT lane's true last samples, D(12,4),
//        D(40,2), PCM.
//
// A run (consecutive blocks of one channel) is one stream: positions count from the run's first
// sample, blocks only matter for the squelch magnitude and the cross-block check values.  A run
// that does not start the call re-derives kFHal samples of history like the other kernels do and
// is checked by finish_channel (y at block-relative position -705).  The kernel finishes its own
// channels: the last wave of a channel's last workgroup runs finish_channel -- from LDS, without a
// memory round trip, when the channel was one workgroup's (`local`).
//
// Mirrors IqDataProcessor::reduceSampleRate (IqDataProcessor.cc:429-500), upconvertByFsOver4
// (:771-815), SignalDetector::detectSignal (SignalDetector.cc:205-274),
// WbFmDemodulator::demodulateSignal / createPcmData (WbFmDemodulator.cc:381-500),
// IirFilter::filterData (IirFilter.cc:161-176), Decimator_int16::filterData (Decimator_int16.cc:176-249).
#ifndef HRFD_FLOW_RING
#define HRFD_FLOW_RING 384
#endif
#ifndef HRFD_FLOW_SVC
#define HRFD_FLOW_SVC 4
#endif
#ifndef HRFD_FLOW_WARM_TILES
#define HRFD_FLOW_WARM_TILES 2      /* warm-up of the recurrence tiles, in tiles of 64 samples (2: ~7e-4 of the tiles are repaired in place) */
#endif
#ifndef HRFD_FLOW_SVC_PRIO
#define HRFD_FLOW_SVC_PRIO 3
#endif
// Round 5, the RE-SPLIT of the WBFM chain: the ring holds the mixed 256 kS/s samples as 16-bit (q, i)
// pairs instead of the float v (a third of the LDS per sample), theta / wrap / numerator move from the stream waves to
// the service waves, which keep a tile's v in REGISTERS through the partial sum, the warm-up passes and the tile
// itself, and the freed LDS holds a first-QUADRANT atan2 table (theta_quad) and a ring of 512 tiles.
// -DHRFD_FLOW_SPLIT=0 builds the round-4 kernel (the A/B of profiles/r5_flow_split_ab.txt).
#ifndef HRFD_FLOW_SPLIT
#define HRFD_FLOW_SPLIT 1
#endif
#ifndef HRFD_FLOW_SVC_WB
#define HRFD_FLOW_SVC_WB 6          /* service waves of the re-split WBFM kernel (they do a third of the work now) */
#endif
// MEASURED AS NOTHING and left off: the stream waves requesting their first unit BEFORE the workgroup's barriers and state
// loads (two memory round trips to the first sample instead of five: profiles/r5_flow_early_ab_NOTHING.txt, 0.2061 against
// 0.2065 ms at 256 channels, 0.7984 against 0.7967 at 1024, alternating runs) -- the service waves have nothing to do for
// the first generation's 3 us either way.  The waves' FIRST units are fixed (wave - SVC) in both builds.
#ifndef HRFD_FLOW_EARLY
#define HRFD_FLOW_EARLY 0
#endif
// The stream's LAST generation: behind the last sample nothing else runs on the CU and the generation's wave makes its
// 64 samples of theta per lane alone (3.5 us at a lone wave's rate).  Theta is a function of the sample alone (no
// neighbour), so the STREAM waves -- ten of them, each with 8 samples per lane of a unit -- make it for the last
// generation's units as they pass through and put the floats into a scratch area of the ring (rows that nobody reads or
// writes any more: the same place HRFD_FLOW_COOP uses); the generation's wave takes them from there and is left with
// the wrap and the numerator.  Same operations on the same operands: same bits.
// MEASURED AND SWITCHED OFF (profiles/r5_flow_lasttheta_ab_LOSES.txt, alternating runs): the service tail does fall, from
// 7.9 to 6.8 us, but 256 channels take the same time (0.2065 against 0.2062 ms) and 1024 channels 1.2 % MORE (0.8047
// against 0.7952).  Bit-exact, the GPU suite and the stress build pass with it on.
#ifndef HRFD_FLOW_LASTTHETA
#define HRFD_FLOW_LASTTHETA 0
#endif
#ifndef HRFD_FLOW_FINISH_FIRST
#define HRFD_FLOW_FINISH_FIRST 1    /* a completed block is finished in front of the iteration's waits (0: behind the ring-space wait, rounds 2-4) */
#endif
#ifndef HRFD_FLOW_RING14
#define HRFD_FLOW_RING14 384        /* AM / SSB: tiles of their ring (512 fits their LDS: the A/B of profiles/r5_fir_ring_ab.txt) */
#endif
#ifndef HRFD_FLOW_RING2
#define HRFD_FLOW_RING2 512         /* its ring, in tiles of 64 samples (a power of two; 256: -1.8 %, profiles/r5_flow_split_ab_4_prio_ring.txt) */
#endif
#ifndef HRFD_FLOW_SVC_PRIO_WB
#define HRFD_FLOW_SVC_PRIO_WB 0     /* s_setprio of its service waves: with a third of the work they need no head start (3: -0.4 % at 1024 channels, same file) */
#endif
// diagnostic build: -DHRFD_FLOW_PROBE accumulates, per stream wave, the cycles between the marks of its unit loop
// (slots 24..31 of its workgroup's stamp row are summed over the waves; read with tools/gpu_flow_times.py)
#ifdef HRFD_FLOW_PROBE
#define FLOW_MARK(i) { const unsigned long long tm_ = __builtin_readcyclecounter(); probe[i] += tm_ - tprev; tprev = tm_; }
#define SVC_MARK(i) { const unsigned long long tm_ = __builtin_readcyclecounter(); sprobe[i] += tm_ - sprev; sprev = tm_; }
// timeline of the workgroup on the 100 MHz constant clock (slots 42..46: entry, tables loaded, last stream wave through,
// last service wave through, last wave at the end); the values of the last launch win
#define FLOW_TIME_SET(i) { if (P.dbg != nullptr && tid == 0) P.dbg[(size_t)blockIdx.x * kDbgSlots + (i)] = __builtin_amdgcn_s_memrealtime(); }
#define FLOW_TIME_MAX(i) { if (P.dbg != nullptr && lane == 0) atomicMax(&P.dbg[(size_t)blockIdx.x * kDbgSlots + (i)], (unsigned long long)__builtin_amdgcn_s_memrealtime()); }
#else
#define FLOW_MARK(i)
#define SVC_MARK(i)
#define FLOW_TIME_SET(i)
#define FLOW_TIME_MAX(i)
#endif

namespace hrfd {

#ifdef HRFD_DUMP_PLAIN_WAITS
constexpr bool kDumpPlainWaits = true;          // TIMING EXPERIMENT ONLY: round 3's counts in the dump builds
#else
constexpr bool kDumpPlainWaits = false;
#endif
constexpr int kFT = 64;                         // samples per tile
constexpr int kFStride = 66;                    // dwords per tile slot: 64-bit accesses of 32 lanes fall into 32 different bank pairs
constexpr int kFRingTiles = HRFD_FLOW_RING;     // tiles of v in the ring
constexpr int kFUnitTiles = 8;                  // a unit = two 4 KiB pieces = 512 samples
constexpr int kFUDw = 1024;                     // U ring: 2048 int16 = two generations (generation g + 2 stores after g + 1 is complete)
constexpr int kFVDw = 256;                      // V ring: 512 int16 = two generations
constexpr int kFEdges = 64;                     // per-unit records kept (>= kFRingTiles / kFUnitTiles + slack)
constexpr int kFPRing = 512;                    // per-tile partial sums kept (eight generations)
constexpr int kFRing2 = HRFD_FLOW_RING2;        // re-split WBFM: tiles of (q, i) pairs in the ring ...
constexpr int kFStride2 = 36;                   // ... 32 dwords each, rows 36 apart: 16 lanes' ds_read_b128 fall into 64 different banks
constexpr int kFEdges2 = 128;                   // ... and unit records (>= kFRing2 / kFUnitTiles + slack)
static_assert((kFRing2 & (kFRing2 - 1)) == 0 && kFRing2 >= 128, "ring of the re-split kernel: a power of two");
static_assert(kFRing2 / kFUnitTiles + 16 <= kFEdges2, "unit records must outlive the ring");
static_assert(kFRingTiles == 256 || kFRingTiles == 320 || kFRingTiles == 384, "ring_slot knows these sizes");
static_assert(kFRingTiles / kFUnitTiles + 16 <= kFEdges, "unit records must outlive the ring");

template <int N>
__device__ __forceinline__ int ring_slot_n(const int t)
{
  static_assert(N == 256 || N == 320 || N == 384 || N == 512, "ring_slot knows these sizes");
  if (N == 256 || N == 512)
  {
    return t & (N - 1);
  }
  const uint32_t h = (uint32_t)t >> 6;                   // t / 64 < 2^15
  const uint32_t q = (N == 320) ? (h * 0xCCCDu) >> 18 : (h * 0xAAABu) >> 18;   // h / 5, h / 6
  return t - (int)q * N;
}
__device__ __forceinline__ int ring_slot(const int t) { return ring_slot_n<kFRingTiles>(t); }

// Hand-offs between the waves of the workgroup go through LDS only.  LDS executes a wave's operations in order
// and has no cache, so "my LDS accesses so far are done" is all a hand-off needs.  A workgroup-scope fence
// would also wait for vmcnt(0): for a stream wave that is the NEXT unit's prefetch -- a full memory latency per
// unit (measured: half of the kernel's time).
__device__ __forceinline__ void lds_order()
{
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
// Every wait of this kernel is bounded: a wave that has spun for ~0.1 s (a protocol error -- the normal waits are
// microseconds) marks its channel as failed (chan_expired), which makes the host replay that channel on the exact
// path, raises the workgroup's ABORT word and leaves its loop; every other wait of the workgroup looks at that word
// every 256 polls and gives up as well, so a stuck workgroup drains within one spin limit, not one per wait.
// `where` identifies the wait in the diagnostics.  RxParams::dbg_flags >> 16 names a wait that workgroup 0 treats as
// expired the first time it polls it (test hook: hrfd_rx_debug_expire).
constexpr uint32_t kFSpinLimit = 1u << 20;
constexpr int kFAbort = 7;                      // ctl[kFAbort] != 0: a wait of this workgroup has expired
struct FlowSpin
{
  uint32_t n = 0;
  __device__ __forceinline__ bool expired(const RxParams &P, uint32_t *ctl, uint32_t &fail_code, uint32_t where)
  {
    ++n;
    const bool forced = (uint32_t)(P.dbg_flags >> 16) == where && blockIdx.x == 0;
    if (n < kFSpinLimit && !forced && ((n & 255u) != 0u || __hip_atomic_load(&ctl[kFAbort], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u))
    {
      return false;
    }
    fail_code = where;                                   // reported as chan_expired at the end of the wave
    __hip_atomic_store(&ctl[kFAbort], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return true;
  }
};
// test hook (hrfd_rx_debug_expire(1000 p + g)): the service wave of generation g of workgroup 0 is held up for ~60 us behind
// hand-over point p of its loop (1: FIR modes, part b handed over; 2: AM / SSB, part c; 3: SSB, 8 kS/s rails published;
// 4: WBFM, partial sums published; 5: WBFM, verification; 6: WBFM, integer stages), while the generations behind it run on;
// 7: a stream wave behind publishing unit g (of every 64); 8: the workgroup of run g in front of its arrival at the channel's
// count (channels cut into several runs: another workgroup finishes the channel then).
// Nothing a held-up wave still has to read may be written over meanwhile: the parity tests run with it.
__device__ __forceinline__ void flow_hold_up(const RxParams &P, const int point, const int g)
{
#ifdef HRFD_FLOW_CHAOS
  // stress build (not shipped): EVERY wave of EVERY workgroup dawdles behind every hand-over point for a pseudo-random
  // 0 .. ~14 us, differently in every launch -- the parity tests and the soak run against it
  {
    uint32_t x = (uint32_t)blockIdx.x * 0x9E3779B1u ^ (uint32_t)g * 0x85EBCA77u ^ (uint32_t)point * 0xC2B2AE3Du ^
                 (uint32_t)__builtin_amdgcn_s_memrealtime();
    x ^= x >> 15;
    x *= 0x2C1B3C6Du;
    x ^= x >> 12;
    if ((x & 3u) == 0u)
    {
      for (uint32_t z = (x >> 8) & 7u; z != 0u; z--)
      {
        __builtin_amdgcn_s_sleep(64);
      }
    }
  }
#endif
  if ((uint32_t)(P.dbg_flags >> 16) == 1000u * (uint32_t)point + (uint32_t)g && blockIdx.x == 0)
  {
    for (int z = 0; z < 15; z++)
    {
      __builtin_amdgcn_s_sleep(127);
    }
  }
}
// A returning LDS atomic whose result is looked at LATER: the compiler would wait for an atomicAdd() at once (and
// LLVM's atomic optimizer puts a readfirstlane right behind it), i.e. for every LDS operation of the wave that is
// still queued in front of it -- hundreds of cycles per unit of a stream wave.  `ret` must go through lds_landed()
// before it is read.  One lane must call this (exec is what the caller's branch left).
__device__ __forceinline__ void lds_add_async(uint32_t &ret, const uint32_t *p, uint32_t val)
{
  const uint32_t off = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t *)p;
  asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(ret) : "v"(off), "v"(val) : "memory");
}
__device__ __forceinline__ void lds_or_async(uint32_t &ret, const uint32_t *p, uint32_t val)
{
  const uint32_t off = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t *)p;
  asm volatile("ds_or_rtn_b32 %0, %1, %2" : "=v"(ret) : "v"(off), "v"(val) : "memory");
}
__device__ __forceinline__ void lds_landed(uint32_t &ret)
{
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ret) : : "memory");
}
__device__ __forceinline__ uint32_t lds_ld(const uint32_t *p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_st(uint32_t *p, uint32_t v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// 64 steps of y <- v - a1*y over one tile of the ring, nothing stored (warm-up)
__device__ __forceinline__ float flow_warm_tile(const uint32_t *tp, float y)
{
  const float a1 = DEEMPH_A1;
  const uint2 *p2 = reinterpret_cast<const uint2 *>(tp);
  uint2 g[32];
#pragma unroll
  for (int j = 0; j < 32; j++)
  {
    g[j] = p2[j];
  }
#pragma unroll
  for (int j = 0; j < 32; j++)
  {
    float r = a1 * y;
    y = u2f(g[j].x) - r;
    r = a1 * y;
    y = u2f(g[j].y) - r;
  }
  return y;
}

// What a lane knows about its tile after running it
struct FlowTile
{
  float y;                       // y of the tile's last sample
  uint32_t sfirst0, sfirst1;     // (int16) y of its first four samples, packed pairs
  uint32_t slast0, slast1;       // ... of its last four samples
  uint32_t ud[8];                // U[0..15] = D(8,4) outputs of the tile, packed pairs; U[0] (low half of ud[0]) is
                                 // filled in later: it needs the left neighbour's last four samples
};

// The tile proper: the recurrence, the (int16_t) narrowing with x86 semantics and D(8,4) in registers.
template <bool FIX>
__device__ __forceinline__ void flow_tile_u(const uint32_t *tp, float y, FlowTile &o)
{
  const float a1 = DEEMPH_A1;
  const uint2 *p2 = reinterpret_cast<const uint2 *>(tp);
  uint2 g[32];
#pragma unroll
  for (int j = 0; j < 32; j++)
  {
    g[j] = p2[j];
  }
  uint32_t pa = 0, pb = 0;
#pragma unroll
  for (int i = 0; i < 16; i++)
  {
    float r = a1 * y;
    const float y0 = u2f(g[2 * i].x) - r;
    r = a1 * y0;
    const float y1 = u2f(g[2 * i].y) - r;
    r = a1 * y1;
    const float y2 = u2f(g[2 * i + 1].x) - r;
    r = a1 * y2;
    y = u2f(g[2 * i + 1].y) - r;
    const uint32_t sa = pack_s16<FIX>(y0, y1), sb = pack_s16<FIX>(y2, y);
    if (i == 0)
    {
      o.sfirst0 = sa;
      o.sfirst1 = sb;
      o.ud[0] = 0u;
    }
    else
    {
      int acc = 1 << 14;
      acc = dot2(pa, kRevWbD1.p[0], acc);
      acc = dot2(pb, kRevWbD1.p[1], acc);
      acc = dot2(sa, kRevWbD1.p[2], acc);
      acc = dot2(sb, kRevWbD1.p[3], acc);
      const uint32_t u16 = (uint32_t)q15_out(acc) & 0xffffu;
      if (i & 1)
      {
        o.ud[i >> 1] = (i == 1) ? (u16 << 16) : (o.ud[i >> 1] | (u16 << 16));
      }
      else
      {
        o.ud[i >> 1] = u16;
      }
    }
    pa = sa;
    pb = sb;
  }
  o.slast0 = pa;
  o.slast1 = pb;
  o.y = y;
}

// The same on a tile whose v the lane holds in REGISTERS (re-split kernel): a warm-up pass over the lane's own tile ...
__device__ __forceinline__ float flow_warm_reg(const float (&v)[kFT], float y)
{
  const float a1 = DEEMPH_A1;
#pragma unroll
  for (int j = 0; j < kFT; j++)
  {
    const float r = a1 * y;
    y = v[j] - r;
  }
  return y;
}

// ... and the tile proper (flow_tile_u on registers)
template <bool FIX>
__device__ __forceinline__ void flow_tile_reg(const float (&v)[kFT], float y, FlowTile &o)
{
  const float a1 = DEEMPH_A1;
  uint32_t pa = 0, pb = 0;
#pragma unroll
  for (int i = 0; i < 16; i++)
  {
    float r = a1 * y;
    const float y0 = v[4 * i] - r;
    r = a1 * y0;
    const float y1 = v[4 * i + 1] - r;
    r = a1 * y1;
    const float y2 = v[4 * i + 2] - r;
    r = a1 * y2;
    y = v[4 * i + 3] - r;
    const uint32_t sa = pack_s16<FIX>(y0, y1), sb = pack_s16<FIX>(y2, y);
    if (i == 0)
    {
      o.sfirst0 = sa;
      o.sfirst1 = sb;
      o.ud[0] = 0u;
    }
    else
    {
      int acc = 1 << 14;
      acc = dot2(pa, kRevWbD1.p[0], acc);
      acc = dot2(pb, kRevWbD1.p[1], acc);
      acc = dot2(sa, kRevWbD1.p[2], acc);
      acc = dot2(sb, kRevWbD1.p[3], acc);
      const uint32_t u16 = (uint32_t)q15_out(acc) & 0xffffu;
      if (i & 1)
      {
        o.ud[i >> 1] = (i == 1) ? (u16 << 16) : (o.ud[i >> 1] | (u16 << 16));
      }
      else
      {
        o.ud[i >> 1] = u16;
      }
    }
    pa = sa;
    pb = sb;
  }
  o.slast0 = pa;
  o.slast1 = pb;
  o.y = y;
}

constexpr int kFinWords = 168;

// GATED: the exact second pass over channels whose squelch gate closed inside the batch.  The batch launch (GATED ==
// false) speculates every gate open; a channel with a closed gate fails ITS verdict (kFailGate, nothing committed) and
// this variant, launched behind it with one workgroup per channel of its mode, redoes only those channels: the stream of the
// blocks the tracker ALLOWS (Squelch.cc:227-273: present now or in the block before), from the committed state -- the
// demodulator does not see the squelched blocks at all (IqDataProcessor.cc:961-1034: acceptIqData is not called, its
// state is frozen), while the front end runs over everything (a block's 16 bytes of history are its physical
// predecessor's last).  Squelched blocks get zero PCM and n_pcm = 0.  Every other channel's workgroup leaves at once.
// DUMP: the 256 kS/s stream of `enable iqdump` (IqDataProcessor.cc:953-957) goes out as well, 8 bytes per lane and piece.
//
// MODE: 3 WBFM (everything above).  2 (FM) and 14 (AM or SSB, read from the channel's configuration) run the FIR
// demodulators in the same shape -- one persistent workgroup per channel, the whole call as ONE stream: the stream
// waves stop behind the Fs/4 mixer and put the 256 kS/s samples into the ring as two rails of int16 PAIRS (offset
// binary: the first decimator's rounding constant absorbs the 128), the service waves run the channel's decimators on
// tiles straight out of the ring (fir_service_*, below).  Nothing of that has a time-parallel speculation in it: the
// 8 kS/s dc-removal recurrence of AM / SSB runs sequentially, 128 steps per generation, in generation order, so these
// modes need no tail kernel, no verification and no cross-workgroup hand-over.  One unit of history (512 samples)
// sits in front of the stream, filled from the carried input tail.  One run per channel only (the host sees to it).
// LDS of the kernel, as dword offsets into one array (the arrays of the three modes differ in size, and
// k_rx_flow_bank runs all three bodies over one allocation)
template <int MODE>
struct FlowLds
{
  static constexpr bool kSplit = (MODE == 3) && (HRFD_FLOW_SPLIT != 0);   // the re-split WBFM chain (round 5)
  static constexpr bool kAtan = (MODE != 14) && !kSplit;   // theta_tab: FM, and the round-4 WBFM build
  static constexpr int kRails = (MODE == 14) ? 2 : 1;    // AM / SSB keep both rails through all three decimators
  static constexpr int kVDw = (MODE == 14) ? 2 * kFVDw : kFVDw;   // V ring per rail (AM / SSB: four generations, fir service c)
  static constexpr int k8k = (MODE == 14) ? 512 : 4;     // AM / SSB: int16 per rail of the 8 kS/s rings (four generations)
  static constexpr int kRingTiles = kSplit ? kFRing2 : (MODE == 14) ? HRFD_FLOW_RING14 : kFRingTiles;
  static constexpr int kStride = kSplit ? kFStride2 : kFStride;
  static constexpr int kEdges = (kSplit || kRingTiles > 384) ? kFEdges2 : kFEdges;
  static constexpr int oTq = 0;                          // re-split: the first-quadrant table FIRST (its index is the LDS address)
  static constexpr int oRing = kSplit ? kQuadDwords : 0;
  static constexpr int oAtcorr = oRing + kRingTiles * kStride;
  static constexpr int oAtt0 = oAtcorr + (kAtan ? kCorrBytes / 4 : 4);
  static constexpr int oUring = oAtt0 + (kAtan ? kCorrBytes : 4);
  static constexpr int oVring = oUring + kRails * kFUDw;
  static constexpr int oR8k = oVring + kRails * kVDw;
  static constexpr int oXs = oR8k + k8k;                  // (two rails of int16 = k8k dwords)
  static constexpr int oYs = oXs + ((MODE == 14) ? 128 : 4);
  static constexpr int oRcar = oYs + ((MODE == 14) ? 128 : 4);
  static constexpr int oThfin = oRcar + 4;
  static constexpr int oEdges = oThfin + 4;
  static constexpr int oUflag = oEdges + ((kSplit || MODE != 3) ? 4 : 4 * kEdges);   // (unit edges: the round-4 WBFM build only)
  static constexpr int oParr = oUflag + kEdges;
  static constexpr int oPflag = oParr + kFPRing;
  static constexpr int oCtl = oPflag + 8;
  static constexpr int oMagl = oCtl + 32;
  static constexpr int oDbfs = oMagl + 16 * 64;
  static constexpr int oBlkout = oDbfs + 32;
  static constexpr int oWfin = oBlkout + 64;
  static constexpr int oFinl = oWfin + 4;
  static constexpr int oBlist = oFinl + kFinWords;
  static constexpr int oGctl = oBlist + 16;
  static constexpr int oLastdw = oGctl + 4;               // re-split: per generation, the last ring word of its last tile (two samples)
  static constexpr int oWcar = oLastdw + 8;               // ... per warm-up pass and generation: y of its last lane at the end of the pass
  static constexpr int oWflag = oWcar + 4 * 8;            // ... and the flags of those
  static constexpr int kTotal = oWflag + 4 * 8;
  static_assert(kTotal * 4 <= 163840, "LDS");
  static_assert((oRing % 4) == 0 && (oAtcorr % 4) == 0 && (oAtt0 % 4) == 0 && (oUring % 4) == 0 && (oVring % 4) == 0 && (oXs % 4) == 0 && (oYs % 4) == 0 &&
                (kFinWords % 4) == 0, "16-byte alignment of what is accessed as 128-bit words");
};
constexpr int kCtlRel = 24;                     // ctl[]: re-split, generations whose ring rows have been read (released to the stream waves)
constexpr int kCtlBlk = 29;                     // ctl[29], ctl[30]: one bit per block of the run (<= 64): finished (its magnitude slot and unit count are free again)
constexpr int kCtlTab = 25;                     // ... service waves that have copied their share of the table
// The stream's LAST generation is made together: behind the last sample nothing else runs on the CU, and a lane's 64
// samples of theta / wrap / numerator are 3.5 us of one wave while five service waves idle.  Its tiles are cut into
// four QUARTERS of 16 samples; the generation's wave and every service wave that has run out of generations claim
// quarters (a bit each in ctl[kCtlCoopNext]), put their v into a scratch area of the ring (rows that nobody reads or writes any
// more) and the generation's wave collects its 64 per lane from there.  Same operations on the same operands: same bits.
// MEASURED AND SWITCHED OFF (profiles/r5_flow_coop_ab_LOSES.txt, alternating runs on one box): 0.2094 against 0.2081 ms
// at 256 channels, 0.7978 against 0.7965 at 1024 -- the helpers' claims, the second pass over LDS and four separate
// stretches (each re-deriving the two samples in front of it) cost what the shared work saves; the service tail stays
// at 7.8 us.  Kept as a build flag (bit-exact, the GPU suite and the stress build pass with it on).
#ifndef HRFD_FLOW_COOP
#define HRFD_FLOW_COOP 0
#endif
constexpr int kCtlCoopOpen = 26;                // ... the last generation's rows may be read (its units are in the ring): g + 1
constexpr int kCtlCoopNext = 27;                // ... the quarters that are taken, one bit each
constexpr int kCtlCoopDone = 28;                // ... quarters finished
constexpr int kCoopStride = 68;                 // dwords per tile in the scratch area: 16 lanes' 128-bit accesses fall into 64 different banks
constexpr int kCoopRows = (64 * kCoopStride + 128 + kFStride2 - 1) / kFStride2;   // ring rows of the scratch area (v, then theta and b0*x of every tile's last sample)
static_assert(384 + kCoopRows <= kFRing2 || kFRing2 < 512, "the scratch area fits beside the last generation's rows");

// The flow kernel as an object: what the sections below share -- the workgroup's LDS arrays, the channel, the run's
// geometry, the carried state, the failure code -- are its members, set once by setup(); the sections are member
// functions, each readable by itself:
//   setup()                 map the workgroup to (channel, run), GATED: build the block list; tables, flags and carried
//                           pipelines into LDS (the kernel's only two workgroup barriers)
//   stream_waves()          waves SVC .. 15: raw IQ -> the ring (phase A; every mode)
//   service_waves_fir()     waves 0 .. SVC-1, FM and AM / SSB: rails -> decimators -> PCM
//   service_waves_wbfm()    waves 0 .. SVC-1, WBFM: v -> recurrence tiles -> integer stages -> PCM
//   finish()                the channel's verdict and commit (from LDS when the channel was this workgroup's alone)
// Everything is inlined into the kernel; the object is scalarised by the compiler (no scratch: tools/kinfo.sh).
template <int SVC, bool GATED, bool DUMP, int MODE>
struct Flow
{
  static_assert(MODE == 3 || MODE == 2 || MODE == 14, "WBFM, FM, AM / SSB");
  typedef FlowLds<MODE> Lds;
  static constexpr bool kWb = (MODE == 3);
  static constexpr bool kSplit = Lds::kSplit;            // WBFM, re-split (round 5): (q, i) pairs in the ring, theta in the service waves
  static constexpr bool kAtan = Lds::kAtan;
  static constexpr int kRails = Lds::kRails;
  static constexpr int kVDw = Lds::kVDw;

  const RxParams &P;
  uint32_t *const lds;
  // LDS arrays (dword offsets: FlowLds<MODE>)
  uint32_t *ring;
  uint8_t *atcorr;
  float *att0;
  uint32_t *uring, *vring;
  int16_t (*r8k)[Lds::k8k];
  float *xs8k, *ys8k, *rcar;
  uint32_t *thfin;
  uint32_t (*edges)[4];
  uint32_t *uflag;
  float *parr;
  uint32_t *pflag, *ctl;
  uint32_t (*magl)[64];
  int8_t *dbfs8;
  uint32_t *blkout, *wfin, *finl;
  uint8_t *blist;
  uint32_t *gctl;
  uint32_t *tquad, *lastdw, *wflag;
  float *wcar;
  // the workgroup's place: channel, run, lane
  uint32_t ci, run, c;
  int n256, tid, lane, wave;
  uint32_t n_stream_blocks, b_first, b_end;
  bool first, local;
  int hal, L, n_units, n_tiles, n_gens, upb, wt, M;
  const ChanState *st;
  ChanState *so;
  ChanCfg cfg;
  float kgain;
  bool small_y;
  unsigned long long t_kernel, waited;
  uint32_t fail_code;

  __device__ __forceinline__ Flow(const RxParams &P_, uint32_t *const lds_) : P(P_), lds(lds_) {}

  // ------------------------------------------------------------------------------------------------- setup
  __device__ __forceinline__ bool setup()
  {
    ring = lds + Lds::oRing;                  // kFRingTiles * kFStride
    atcorr = reinterpret_cast<uint8_t *>(lds + Lds::oAtcorr);   // theta_tab: correction bytes ...
    att0 = reinterpret_cast<float *>(lds + Lds::oAtt0);           // ... and the first-octant table
    uring = lds + Lds::oUring;                // kRails * kFUDw
    vring = lds + Lds::oVring;                // kRails * kVDw
    // AM / SSB at 8 kS/s: four generations of the last decimator's outputs per rail (SSB: the Hilbert transformer reads 30
    // back), a generation's recurrence input and output, and the recurrence's carried x[n-1], y[n-1]
    r8k = reinterpret_cast<int16_t (*)[Lds::k8k]>(lds + Lds::oR8k);
    xs8k = reinterpret_cast<float *>(lds + Lds::oXs), ys8k = reinterpret_cast<float *>(lds + Lds::oYs);
    rcar = reinterpret_cast<float *>(lds + Lds::oRcar);
    thfin = lds + Lds::oThfin;                // FM: theta of the last four 64 kS/s samples of the last finished generation
    edges = reinterpret_cast<uint32_t (*)[4]>(lds + Lds::oEdges);   // per unit: theta of its first two and last two samples
    uflag = lds + Lds::oUflag;                // unit u is complete in the ring: u + 1
    parr = reinterpret_cast<float *>(lds + Lds::oParr);   // per tile: geometric partial sum of v
    pflag = lds + Lds::oPflag;                // partial sums of generation g are in parr: g + 1
    ctl = lds + Lds::oCtl;                    // 0 next unit, 1 generations verified (their v is released), 2 generations complete (U, V, PCM), 3 blocks finished,
                                                            // 4 next generation, 5 waves of the workgroup that are through,
                                                            // 8..23 units done (per block, mod 16)
    magl = reinterpret_cast<uint32_t (*)[64]>(lds + Lds::oMagl);   // per block (mod 16: a wave with a unit of block
                                             // b >= 16 waits for block b - 16 to be finished, kCtlBlk -- blocks of the
                                             // shortest size are 4 units, sixteen of them no more than the ring spans)
                                             // and lane: sum of the sample magnitudes.  One word per lane: a
                                             // same-address atomic from 64 lanes becomes a 64-step scalar loop (LLVM's atomic
                                             // optimizer), measured at half of the kernel's time
    dbfs8 = reinterpret_cast<int8_t *>(lds + Lds::oDbfs);   // the reachable part of the dBFS table
    blkout = lds + Lds::oBlkout;              // per block of the run: mean magnitude | present << 31 (written out at the end:
                                             // a global store inside the unit loop costs the loop its counted vmcnt waits)
    wfin = lds + Lds::oWfin;                  // the last finished generation's last lane: y, its last two S pairs
    finl = lds + Lds::oFinl;                  // a channel that is ONE workgroup's is finished from here (no memory round trip behind
                                             // the last sample): 0..63 y in front of block b, 128..159 the pending WBFM state
                                             // section, 160 tracking, 161 poison, 162..165 the pending fe_tail
    blist = reinterpret_cast<uint8_t *>(lds + Lds::oBlist);   // GATED: the blocks of the stream, in order (the allowed ones)
    gctl = lds + Lds::oGctl;                  // GATED: 0 number of allowed blocks, 1 `present` of the call's last block; AM / SSB: 2 generations through their 8 kS/s part
    tquad = lds + Lds::oTq;                   // re-split: the first-quadrant atan2 table (theta_quad), copied by the service waves
    lastdw = lds + Lds::oLastdw;              // ... [8] per generation: the last ring word of its last tile
    wcar = reinterpret_cast<float *>(lds + Lds::oWcar);   // ... [4][8] per warm-up pass and generation: its last lane's y behind the pass
    wflag = lds + Lds::oWflag;                // ... [4][8] generation + 1 when that value is there

    if (!map_unit(blockIdx.x, P.n_list, P.n_runs, ci, run))
    {
      return false;
    }
#ifdef HRFD_FLOW_XCDSWAP
    // EXPERIMENT ONLY (profiles/r5_xcd_swap_experiment.txt): does the 4-5 us by which the XCDs with odd numbers trail the
    // even ones follow the XCD or the channel's place in memory?  Neighbouring channels change places.
    if ((ci ^ 1u) < P.n_list)
    {
      ci ^= 1u;
    }
#endif
    c = P.chan_list[ci];
    n256 = (int)P.n256;
    tid = threadIdx.x;
    lane = tid & 63;
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    n_stream_blocks = 0;                          // GATED: blocks of the stream
    if (GATED)
    {
      // only the channels whose one and only failure in the batch launch was a closed gate (the host launches this
      // variant with one run per channel and at most 64 blocks)
      if (P.fin.chan_fail[c] != kFailGate)
      {
        return false;
      }
      if (wave == 0)
      {
        const uint32_t nb = P.n_blocks;
        const uint32_t pres = ((uint32_t)lane < nb) ? (uint32_t)(P.present[(size_t)c * nb + lane] != 0) : 0u;
        const uint32_t prev = shr1(pres, (P.state[c].tracking != 0) ? 1u : 0u);
        const bool allowed = (uint32_t)lane < nb && (pres | prev) != 0u;
        const unsigned long long m = __ballot(allowed);
        if (allowed)
        {
          blist[__popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)lane;
        }
        if (lane == 0)
        {
          gctl[0] = (uint32_t)__popcll(m);
          gctl[1] = (uint32_t)__builtin_amdgcn_readlane((int)pres, (int)nb - 1);
        }
        // squelched blocks hand back silence, not what the batch launch left there
        const uint32_t npcm2 = (uint32_t)n256 >> 6;        // PCM pairs per block
        uint32_t *pz = reinterpret_cast<uint32_t *>(P.pcm + ((size_t)c * P.out_blocks + P.out_b0) * (size_t)(n256 >> 5));
        for (uint32_t b = 0; b < nb; b++)
        {
          if (!((m >> b) & 1ull))
          {
            for (uint32_t i = (uint32_t)lane; i < npcm2; i += 64u)
            {
              pz[(size_t)b * npcm2 + i] = 0u;
            }
          }
        }
      }
      __syncthreads();
      n_stream_blocks = gctl[0];
    }
    b_first = GATED ? 0u : run * P.run_len;
    b_end = GATED ? n_stream_blocks : min(P.n_blocks, b_first + P.run_len);
    first = (b_first == 0);                     // the stream continues from the carried state: exact start
    local = GATED || (kWb && P.self_finish != 0 && P.n_runs == 1u && P.n_blocks <= 64u);   // the channel is this workgroup's alone
    // history in front of the run (samples): WBFM re-derives it from the input when the run does not start the call; the
    // FIR modes always have one unit, from the carried tail (their runs start the call)
    hal = kWb ? (first ? 0 : P.flow_hal) : 512;
    L = hal + (int)(b_end - b_first) * n256;     // samples of the stream
    n_units = L >> 9, n_tiles = L >> 6, n_gens = (n_tiles + 63) >> 6;
    upb = n256 >> 9;                             // units per block
    wt = P.warm_tiles, M = P.seed_terms;
    st = P.state + c;
    so = P.state_out + c;
    t_kernel = __builtin_readcyclecounter();
    waited = 0;
    fail_code = 0;                                // which wait expired, if any (diagnostics)
    return true;
  }

  // The second half of the set-up: the channel's configuration, control words and carried pipelines into LDS, and the
  // kernel's only two workgroup barriers.  Every wave calls it exactly once: the service waves at their start, the stream
  // waves in front of (or, -DHRFD_FLOW_EARLY=1, behind) the requests for their first unit.
  // (the call sites differ between the two kinds of wave, which is outside what HIP promises for
  //  __syncthreads() -- a barrier reached through different code paths.  It holds on gfx950 because s_barrier counts WAVES,
  //  not call sites: the workgroup's barrier is released when every wave of the workgroup has executed one s_barrier,
  //  whichever instruction address it sits at.  What must stay true, and is the whole contract here: every one of the 16
  //  waves executes EXACTLY the two barriers of this function (GATED: and the one of setup() in front of it, which all
  //  waves reach at the same place) and no other, no wave leaves the kernel in front of them
  //  (the early exits of the body come behind setup_lds or are taken by the whole workgroup), and nothing between the
  //  two barriers depends on the kind of wave.  A third barrier anywhere in Flow<> breaks it.)
  __device__ __forceinline__ void setup_lds()
  {
    cfg = P.cfg[c];
    kgain = cfg.gain_wbfm / 75000.0f;                // K = (gain/75000)*32767 in float, that order (WbFmDemodulator.cc:392-395)
    kgain = kgain * 32767.0f;
    small_y = fabsf(kgain) * 3.3f < 2147483000.0f;   // |y| <= |K| pi: the int32 cast cannot overflow
    FLOW_TIME_SET(42)

    // tables and control words
    magl[0][tid] = 0u;
    if (kAtan)
    {
      for (int i = tid; i < kCorrBytes / 4; i += kThreads)
      {
        reinterpret_cast<uint4 *>(att0)[i] = reinterpret_cast<const uint4 *>(P.at_t0)[i];
      }
    }
    if (kAtan && tid < kCorrBytes / 16)
    {
      reinterpret_cast<uint4 *>(atcorr)[tid] = reinterpret_cast<const uint4 *>(P.at_corr2)[tid];
    }
    else if (tid >= 640 && tid < 640 + Lds::kEdges)
    {
      uflag[tid - 640] = 0u;
    }
    else if (tid >= 768 && tid < 800)
    {
      // (ctl[0], the next unit: the stream waves' FIRST units are fixed -- wave - SVC, requested before this barrier --
      //  and the counter starts behind them; GATED: everything is taken from the counter)
      ctl[tid - 768] = (tid == 768 && kWb && !GATED) ? (uint32_t)(kWaves - SVC) : 0u;
    }

    else if (tid >= 800 && tid < 808)
    {
      pflag[tid - 800] = 0u;
    }
    else if (tid >= 840 && tid < 872)
    {
      reinterpret_cast<uint32_t *>(dbfs8)[tid - 840] = 0u;
    }
    else if (kWb && tid >= 832 && tid < 836 && first)
    {
      // what the lane in front of tile 0 would have left: the carried y and the last four S samples
      wfin[tid - 832] = (tid == 832) ? f2u(st->wb_y) : (tid == 833) ? reinterpret_cast<const uint32_t *>(st->wb_s)[0]
                                                     : (tid == 834) ? reinterpret_cast<const uint32_t *>(st->wb_s)[1] : 0u;
    }
    else if (kWb && tid >= 896 && tid < 900 && first)
    {
      uring[kFUDw - 4 + (tid - 896)] = reinterpret_cast<const uint32_t *>(st->wb_u)[tid - 896];     // U[-8 .. -1]
    }
    else if (tid == 904 && local)
    {
      finl[160] = st->tracking;
      finl[161] = P.fin.chan_poison[c];
    }
    else if (kSplit && tid >= 980 && tid < 1012)
    {
      wflag[tid - 980] = 0u;
    }
    else if (tid >= 908 && tid < 912 && (GATED || b_end == P.n_blocks))
    {
      // front-end carry for the next call: the last 16 raw bytes of the channel's input (pending, like the rest of state_out)
      const int8_t *endp = P.iq + (uint64_t)c * P.ch_stride + (uint64_t)P.n_blocks * P.block_bytes;
      const uint32_t w = reinterpret_cast<const uint32_t *>(endp - 16)[tid - 908];
      reinterpret_cast<uint32_t *>(so->fe_tail)[tid - 908] = w;
      finl[162 + (tid - 908)] = w;
    }
    else if (kWb && tid >= 960 && tid < 979 && first)
    {
      vring[kFVDw - 19 + (tid - 960)] = reinterpret_cast<const uint32_t *>(st->wb_v)[tid - 960];    // V[-38 .. -1]
    }
    __syncthreads();
    if (!kWb)
    {
      // The FIR modes' stream begins with one unit of HISTORY (tiles 0 .. 7 = positions -512 .. -1): the tail of the
      // 256 kS/s stream this demodulator consumed last (ChanState: offset-binary bytes i, q per sample -- the ring's own
      // number format), zeros (0x80) in front of it.  The decimators warm up over it (they reach 260 samples back,
      // FM's tuner 28); what they produce there is discarded.  Stage pipelines that carry samples scaled with the gain
      // of their time (FM: U, V) and the 8 kS/s histories come from the state instead, as in the reference's objects.
      const bool am = (MODE == 14) && cfg.mode == 1;
      const uint8_t *tail = (MODE == 2) ? st->fm_tail + 2 * (kFmTail - 512) : am ? st->am_tail : st->ssb_tail;
      constexpr int kHave = (MODE == 2) ? 512 : kAmTail;   // samples of history the state holds (of the 512)
      if (tid < 256)
      {
        // pair k = samples 2k, 2k + 1 of the history unit
        const int k0 = tid - (512 - kHave) / 2;
        uint32_t ip = 0x00800080u, qp = 0x00800080u;
        if (k0 >= 0)
        {
          const uint32_t w = reinterpret_cast<const uint32_t *>(tail)[k0];   // i0 q0 i1 q1
          ip = (w & 0xffu) | ((w & 0x00ff0000u));
          qp = ((w >> 8) & 0xffu) | ((w >> 8) & 0x00ff0000u);
        }
        ring[(tid >> 5) * kFStride + (tid & 31)] = ip;
        ring[(tid >> 5) * kFStride + 32 + (tid & 31)] = qp;
      }
      else if (MODE == 2 && tid >= 256 && tid < 260)
      {
        uring[8 * 8 - 4 + (tid - 256)] = reinterpret_cast<const uint32_t *>(st->fm_u)[tid - 256];          // U[-8 .. -1] (tile 8 is position 0)
      }
      else if (MODE == 2 && tid >= 320 && tid < 339)
      {
        vring[(2 * 8 - 19 + (tid - 320)) & (kFVDw - 1)] = reinterpret_cast<const uint32_t *>(st->fm_v)[tid - 320];   // V[-38 .. -1]
      }
      else if (MODE == 14 && tid >= 384 && tid < 400 && !am)
      {
        // SSB: the last 32 samples of the 8 kS/s rails in front of sample 0 (index 16 here: the history unit yields 16)
        reinterpret_cast<uint32_t *>(r8k[0])[(tid - 384 + 248) & 255] = reinterpret_cast<const uint32_t *>(st->ssb_i)[tid - 384];
        reinterpret_cast<uint32_t *>(r8k[1])[(tid - 384 + 248) & 255] = reinterpret_cast<const uint32_t *>(st->ssb_q)[tid - 384];
      }
      else if (MODE == 14 && tid == 448)
      {
        rcar[0] = am ? st->am_x1 : st->ssb_x1;
        rcar[1] = am ? st->am_y1 : st->ssb_y1;
      }
      else if (tid == 512)
      {
        gctl[2] = 0u;                                       // AM / SSB: generations through their 8 kS/s recurrence
        gctl[3] = 0u;                                       // SSB: generations whose 8 kS/s rails are in their rings
        ctl[0] = 1u + (GATED ? 0u : (uint32_t)(kWaves - SVC));   // the stream starts with unit 1; the waves' first units are fixed (see above)
        uflag[0] = 1u;                                      // unit 0, the history, is in the ring
        thfin[0] = thfin[1] = thfin[2] = thfin[3] = 0u;
      }
    }
    if (tid < 128)
    {
      dbfs8[tid] = (int8_t)P.dbfs[tid];
    }
    __syncthreads();                                       // the only workgroup barriers of the kernel
    FLOW_TIME_SET(43)
  }

  // ------------------------------------------------------------------------------------------ stream waves
  __device__ __forceinline__ void stream_waves()
  {
    // =================================================================== stream waves: raw IQ -> v
    StreamCtx X;
    X.P = &P;
    X.atc = atcorr;
    X.ati = att0;
    X.lane = lane;
    const uint32_t base_off = b_first * P.block_bytes - (uint32_t)hal * 16u;   // byte offset of stream sample 0
    const uint32_t dead = 0xffff0000u;                   // outside the descriptor: zeros, no memory traffic
    // byte offset of unit u's input.  GATED: the stream is a list of blocks (units are taken in ascending order, so a
    // wave follows the list with a cursor)
    int cur_bu0 = hal >> 9, cur_blk = 0;                  // (the history unit(s) in front of the stream belong to no block)
    auto unit_off = [&](const int uu) -> uint32_t {
      if (!GATED)
      {
        return base_off + (uint32_t)uu * 8192u;
      }
      while (uu >= cur_bu0 + (n256 >> 9))
      {
        cur_bu0 += n256 >> 9;
        cur_blk++;
      }
      return (uint32_t)blist[cur_blk & 63] * P.block_bytes + (uint32_t)(uu - cur_bu0) * 8192u;
    };
    auto grab = [&]() -> int {
      uint32_t g = 0;
      if (lane == 0)
      {
        g = atomicAdd(&ctl[0], 1u);
      }
      return __builtin_amdgcn_readfirstlane((int)g);
    };
    // The raw loads of the unit loop are hand-placed: inline asm with the destination tied to the registers the loop
    // already owns ("+v"), so that an in-flight load never needs a copy at the back edge (a copy reads the register:
    // s_waitcnt vmcnt(0) once per unit, which is what the compiler made of builtin loads here), and counted waits
    // (vm_wait).  The compiler sees no vector memory operation in the loop and adds no waits of its own.
    // In flight, oldest first, at the top of every iteration: c16 (1), qa (4), qb (4).
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const uint64_t basep = (uint64_t)(P.iq + (uint64_t)c * P.ch_stride);
    i32x4 desc;
    desc.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)basep);
    desc.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(basep >> 32) & 0xffffu));
    desc.z = __builtin_amdgcn_readfirstlane((int)(P.n_blocks * P.block_bytes));
    desc.w = 0x00020000;
    const int lane_off = lane * 64;
    // DUMP: the run's part of the 256 kS/s output, 2 bytes per sample, as a second buffer (lane l of a piece holds
    // samples 4l .. 4l + 3: eight contiguous bytes)
    i32x4 ddesc = desc;
    const int lane_off8 = lane * 8;
    if (DUMP)
    {
      const uint64_t dbase = (uint64_t)(P.iq256 + ((size_t)c * P.out_blocks + P.out_b0 + b_first) * (size_t)(2 * n256));
      ddesc.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)dbase);
      ddesc.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(dbase >> 32) & 0xffffu));
      ddesc.z = __builtin_amdgcn_readfirstlane((int)((b_end - b_first) * (uint32_t)(2 * n256)));
    }
    auto store_dump = [&](const uint32_t (&w)[2], const int uu, const int half) {
      // stream sample 512 uu + 256 half (+ 4 lane) minus the history in front of the run; history is not dumped
      const int pos = 512 * uu + 256 * half - hal;
      const uint32_t soff = (pos >= 0) ? (uint32_t)(2 * pos) : dead;
      typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
      const u32x2 d = {w[0], w[1]};
      asm volatile("buffer_store_dwordx2 %0, %1, %2, %3 offen" : : "v"(d), "v"(lane_off8), "s"(ddesc), "s"(soff) : "memory");
    };
    // (a dump store outside the descriptor: no memory traffic, but a vector memory operation like the real ones)
    auto store_nothing = [&]() {
      typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
      const u32x2 d = {0u, 0u};
      asm volatile("buffer_store_dwordx2 %0, %1, %2, %3 offen" : : "v"(d), "v"(lane_off8), "s"(ddesc), "s"(dead) : "memory");
    };
    // `dep`: stage-1 outputs that between them have read every raw register of the piece being refilled -- the asm
    // names them as inputs, so no read of the old contents can be scheduled behind the load
    auto load_piece = [&](u32x4 (&q)[4], const uint32_t uoff, const int half, const bool live, const uint32_t (&dep)[4][4]) {
      const uint32_t soff = live ? uoff + (uint32_t)half * 4096u : dead;
#if (HRFD_ABLATE & 128)
      for (int j = 0; j < 4; j++)
      {
        q[j] = u32x4{lane * 0x01010101u + half, lane * 0x3010501u + j, half * 0x10101u, lane ^ (half + soff)};   // TIMING EXPERIMENT ONLY: no HBM reads
      }
      return;
#endif
      asm volatile("buffer_load_dwordx4 %0, %4, %5, %6 offen\n\t"
                   "buffer_load_dwordx4 %1, %4, %5, %6 offen offset:16\n\t"
                   "buffer_load_dwordx4 %2, %4, %5, %6 offen offset:32\n\t"
                   "buffer_load_dwordx4 %3, %4, %5, %6 offen offset:48"
                   : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3])
                   : "v"(lane_off), "s"(desc), "s"(soff), "v"(dep[0][1]), "v"(dep[0][3]), "v"(dep[1][1]), "v"(dep[1][3]),
                     "v"(dep[2][1]), "v"(dep[2][3]), "v"(dep[3][1]), "v"(dep[3][3])
                   : "memory");
    };
    // the 16 bytes in front of a unit: its front-end carries depend on nothing else
    // (uoff == 0: the call's very first bytes -- what is in front of them is the carried fe_tail)
    auto load_c16 = [&](u32x4 &q, const uint32_t uoff, const bool live) {
      const uint32_t soff = (live && !(first && uoff == 0u)) ? uoff - 16u : dead;
#if (HRFD_ABLATE & 128)
      q = u32x4{0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
      return;
#endif
      asm volatile("buffer_load_dwordx4 %0, off, %1, %2" : "+v"(q) : "s"(desc), "s"(soff) : "memory");
    };
#if (HRFD_ABLATE & 128)
#define VM_WAIT(n, ...)
#else
#define VM_WAIT(n, ...) asm volatile("s_waitcnt vmcnt(" #n ")" : __VA_ARGS__ : : "memory")
#endif
    // lane constant: where this lane's four samples of a piece go (tile lane / 16 of the piece, 4 (lane % 16) inside)
    const int lane_dw = kSplit ? kFStride2 * (lane >> 4) + ((2 * lane) & 31)   // re-split: one dword per PAIR of samples, bytes i0 q0 i1 q1
                        : kWb  ? kFStride * (lane >> 4) + ((4 * lane) & 63)    // one dword per sample
                               : kFStride * (lane >> 4) + ((2 * lane) & 31);   // FIR modes: one dword per PAIR, I rail at 0, Q rail at 32
    int bu0 = hal >> 9, blk = 0;                         // first unit and index (in the run) of the block a unit belongs to
    bool scratch_ok = false;                             // (HRFD_FLOW_LASTTHETA: the scratch area and the table are there)
    unsigned long long probe[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
    (void)probe; (void)tprev;
    // the unit that completes a block (its count came back as upb - 1) finishes the block: block-mean magnitude,
    // detector (SignalDetector.cc:255, DbfsCalculator.cc:111-147 with a 7-bit full scale).  A batch speculates
    // every gate open; k_rx_epilogue runs the tracker and checks.
    uint32_t pend_nth = 0u;
    bool pend = false;
    int pend_slot = 0, pend_blk = 0;
    auto finish_block = [&]() {
      if (pend)
      {
        lds_landed(pend_nth);
      }
      if (pend && __builtin_amdgcn_readfirstlane((int)pend_nth) == upb - 1)
      {
        uint32_t total = magl[pend_slot][lane];
        magl[pend_slot][lane] = 0u;
        for (int off = 32; off > 0; off >>= 1)
        {
          total += __shfl_down(total, off);
        }
        if (lane == 0)
        {
          const uint32_t mean_mag = total / (uint32_t)n256;
          int32_t dbfs = (int32_t)dbfs8[min(mean_mag, 127u)] - 42;
          dbfs = (int32_t)((uint32_t)dbfs - P.gain_db);
          blkout[pend_blk & 63] = mean_mag | ((dbfs >= cfg.threshold) ? 0x80000000u : 0u);
          lds_st(&ctl[8 + pend_slot], 0u);
          atomicOr(&ctl[kCtlBlk + ((pend_blk >> 5) & 1)], 1u << (pend_blk & 31));   // the slot is block pend_blk + 16's now
          atomicAdd(&ctl[3], 1u);                        // blocks finished
        }
      }
      pend = false;
    };
    // The wave's FIRST unit is fixed (wave - SVC: ctl[0] starts behind these) and requested BEFORE the workgroup's barriers
    // and the state's loads (setup_lds, below): a workgroup's first raw bytes are in flight one memory round trip into
    // its life.  GATED: the units come from the list of allowed blocks, which setup() has built; everything from the counter.
    if (GATED || !HRFD_FLOW_EARLY)
    {
      setup_lds();
    }
    int u = GATED ? grab() : (kWb ? 0 : 1) + (wave - SVC);
    u32x4 qa[4], qb[4], c16;
    for (int j = 0; j < 4; j++)
    {
      qa[j] = u32x4{0u, 0u, 0u, 0u};
      qb[j] = u32x4{0u, 0u, 0u, 0u};
    }
    c16 = u32x4{0u, 0u, 0u, 0u};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // nothing of the prologue's is in flight: the counts below are exact
    uint32_t uoff = (u < n_units) ? unit_off(u) : 0u, uoff_n = 0u;   // byte offsets of unit u and of the next one
    load_c16(c16, uoff, u < n_units);
    {
      const uint32_t nodep[4][4] = {};
      load_piece(qa, uoff, 0, u < n_units, nodep);
      if (DUMP)
      {
        store_nothing();                                 // (where a unit's first dump store stands among its loads: the counts below)
      }
      load_piece(qb, uoff, 1, u < n_units, nodep);
      if (DUMP)
      {
        store_nothing();
      }
    }
    // the carried state of a stream that continues the previous call: requested behind the unit, looked at behind the barriers
    const float th_v = st->wb_theta, p_v = st->wb_p;
    const uint4 tailv = *reinterpret_cast<const uint4 *>(st->fe_tail);
    if (!GATED && HRFD_FLOW_EARLY)
    {
      setup_lds();
    }
    X.kgain = kgain;
    const uint32_t theta_in = (uint32_t)__builtin_amdgcn_readfirstlane((int)f2u(th_v));
    const uint32_t p_in = (uint32_t)__builtin_amdgcn_readfirstlane((int)f2u(p_v));
    const uint32_t tail_in[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)tailv.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)tailv.y),
                                 (uint32_t)__builtin_amdgcn_readfirstlane((int)tailv.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)tailv.w)};
    while (u < n_units)
    {
      FLOW_MARK(0)
      const uint32_t done_seen = lds_ld(&ctl[kSplit ? kCtlRel : 1]);
      FLOW_MARK(1)
      QuadCarry cy;
      // (DUMP: the two dump stores of a unit stand among the loads, in order: c16' qa' st0 qb' st1 -- behind c16' are 10
      //  operations, behind qa' 6, behind qb 7 (st1, c16', qa', st0); without them 8, 4 and 5.  Round 3 waited with the
      //  plain counts in the dump build as well, i.e. for more than it needed; an A/B on one box (tools/dump_ab.sh,
      //  -DHRFD_DUMP_PLAIN_WAITS) shows no difference: 0.2408 against 0.2394 ms -- what the dump costs is its bytes)
      if (DUMP && !kDumpPlainWaits)
      {
        VM_WAIT(10, "+v"(c16));
      }
      else
      {
        VM_WAIT(8, "+v"(c16));
      }
      {
        const bool st0 = first && uoff == 0u;            // "the 16 bytes in front" of a continued stream are the carried ones
        cy.fe = carry_from_16(make_uint4(st0 ? tail_in[0] : c16.x, st0 ? tail_in[1] : c16.y, st0 ? tail_in[2] : c16.z,
                                         st0 ? tail_in[3] : c16.w));
      }
      // a unit's first two v are provisional (theta and b0*x of the sample in front are another wave's): patched by
      // the service wave -- except at the very start of a stream that continues the previous call
      cy.theta = (first && u == 0) ? theta_in : 0u;
      cy.p = (first && u == 0) ? p_in : 0u;
      FLOW_MARK(2)
      // the previous unit's block, if that unit completed it -- in front of every wait of this iteration: a wave that waits
      // for a block to be finished (below, its magnitude slot) never waits for a wave that is waiting itself
      if (HRFD_FLOW_FINISH_FIRST)
      {
        finish_block();
      }
      // ring space: the tiles this unit overwrites must not be anybody's warm-up any more
      // (re-split: a generation's rows are free as soon as its service wave has READ them -- kCtlRel -- and nobody reads
      //  a row twice: no slack for warm-ups)
      const int ring_slack = kSplit ? 0 : wt;
      if (!(HRFD_ABLATE & (1024 | 2048)) && 8 * u + 8 + ring_slack > 64 * (int)done_seen + Lds::kRingTiles)   // (1024: TIMING EXPERIMENT ONLY, stream waves alone)
      {
        const unsigned long long t0 = __builtin_readcyclecounter();
        FlowSpin sp;
        while (8 * u + 8 + ring_slack > 64 * (int)lds_ld(&ctl[kSplit ? kCtlRel : 1]) + Lds::kRingTiles && !sp.expired(P, ctl, fail_code, 1))
        {
          __builtin_amdgcn_s_sleep(8);
        }
        waited += __builtin_readcyclecounter() - t0;
        if (fail_code != 0u)
        {
          break;                                         // the workgroup is aborting: the channel will be replayed
        }
      }
      // (re-split, HRFD_FLOW_LASTTHETA) the units of the stream's last generation also leave their thetas: in a scratch
      // area whose rows belonged to generations <= n_gens - 6 (read by then: kCtlRel), from the table the service waves
      // have brought (kCtlTab: a stream of one generation gets here before they have)
      const bool lastgen = kSplit && (HRFD_FLOW_LASTTHETA != 0) && u >= 8 * (n_gens - 1);
      if (lastgen && !scratch_ok)
      {
        const unsigned long long t0 = __builtin_readcyclecounter();
        FlowSpin sp;
        while (((int)lds_ld(&ctl[kCtlRel]) < n_gens - 5 || lds_ld(&ctl[kCtlTab]) != (uint32_t)SVC) && !sp.expired(P, ctl, fail_code, 14))
        {
          __builtin_amdgcn_s_sleep(4);
        }
        waited += __builtin_readcyclecounter() - t0;
        lds_order();
        if (fail_code != 0u)
        {
          break;
        }
        scratch_ok = true;
      }
      if (!HRFD_FLOW_FINISH_FIRST)
      {
        finish_block();                                  // (where it stood until round 5)
      }
      FLOW_MARK(3)
      // (uniform, and said so: left to itself the compiler computes the row offset per lane with a quarter-rate v_mul_lo_u32)
      const int slot0 = __builtin_amdgcn_readfirstlane(kSplit ? ((8 * u) & (kFRing2 - 1)) : ring_slot_n<Lds::kRingTiles>(8 * u));   // NT is a multiple of 8: a unit never wraps
      uint32_t *dst = ring + slot0 * Lds::kStride + lane_dw;
      uint32_t v[4], mag4, magsum;
      uint32_t iqb[2] = {0u, 0u};
      float theta[4];
      // the next unit is taken BEFORE piece 0 (the counter's LDS round trip hides behind the piece; taking it behind the
      // piece was measured slower in round 2 and is gone)
      uint32_t un_v = 0;
      if (lane == 0)
      {
        lds_add_async(un_v, &ctl[0], 1u);
      }
      // (the next unit's loads go out from inside the pieces, as soon as a piece's raw registers are free:
      //  almost two pieces of lead without a register more)
      int un = 0;
      if (DUMP && !kDumpPlainWaits)
      {
        VM_WAIT(6, "+v"(qa[0]), "+v"(qa[1]), "+v"(qa[2]), "+v"(qa[3]));
      }
      else
      {
        VM_WAIT(4, "+v"(qa[0]), "+v"(qa[1]), "+v"(qa[2]), "+v"(qa[3]));
      }
      // what happens at the point where a piece's raw registers are free: the next unit's first loads
      auto refill_a = [&](const uint32_t (&y1)[4][4]) {
        lds_landed(un_v);
        un = __builtin_amdgcn_readfirstlane((int)un_v);
        // fairness: the arbiter serves the oldest wave of a SIMD first, so the young ones fall behind, hold the
        // ring's completed frontier back and the old ones run into the ring limit.  A wave that sees more units
        // taken since its own than there are stream waves is late: it gets priority until its next unit.
        {
          const int lag = un - u;
          if (lag > 16)
          {
            __builtin_amdgcn_s_setprio(2);
          }
          else if (lag > 13)
          {
            __builtin_amdgcn_s_setprio(1);
          }
          else
          {
            __builtin_amdgcn_s_setprio(0);
          }
        }
        uoff_n = (un < n_units) ? unit_off(un) : 0u;
        load_c16(c16, uoff_n, un < n_units);
        load_piece(qa, uoff_n, 0, un < n_units, y1);
      };
      // FIR modes: the lane's four mixed samples as int16 pairs per rail (offset binary), two 8-byte stores
      auto store_rails = [&](const uint32_t (&mx)[4], uint32_t *d) {
        const uint32_t i01 = __builtin_amdgcn_perm(mx[1], mx[0], 0x05040100u), i23 = __builtin_amdgcn_perm(mx[3], mx[2], 0x05040100u);
        const uint32_t q01 = __builtin_amdgcn_perm(mx[1], mx[0], 0x07060302u), q23 = __builtin_amdgcn_perm(mx[3], mx[2], 0x07060302u);
        reinterpret_cast<uint2 *>(d)[0] = make_uint2(i01, i23);
        reinterpret_cast<uint2 *>(d + 32)[0] = make_uint2(q01, q23);
        mag4 = magnitude(mx[0]) + magnitude(mx[1]) + magnitude(mx[2]) + magnitude(mx[3]);
      };
      // re-split WBFM: the lane's four mixed samples as two ring words (bytes i0 q0 i1 q1, offset binary), one 8-byte store
      auto store_pairs = [&](const uint32_t (&mx)[4], uint32_t *d) {
        const uint32_t w0 = __builtin_amdgcn_perm(mx[1], mx[0], 0x06040200u), w1 = __builtin_amdgcn_perm(mx[3], mx[2], 0x06040200u);
        reinterpret_cast<uint2 *>(d)[0] = make_uint2(w0, w1);
        mag4 = magnitude(mx[0]) + magnitude(mx[1]) + magnitude(mx[2]) + magnitude(mx[3]);
        if (DUMP)
        {
          iqb[0] = w0 ^ 0x80808080u;
          iqb[1] = w1 ^ 0x80808080u;
        }
      };
      if (kSplit)
      {
        const uint4 ra[4] = {make_uint4(qa[0].x, qa[0].y, qa[0].z, qa[0].w), make_uint4(qa[1].x, qa[1].y, qa[1].z, qa[1].w),
                             make_uint4(qa[2].x, qa[2].y, qa[2].z, qa[2].w), make_uint4(qa[3].x, qa[3].y, qa[3].z, qa[3].w)};
        uint32_t mx[4];
        quad_front(ra, cy.fe, mx, refill_a);
        store_pairs(mx, dst);
      }
      else if (kWb)
      {
        const uint4 ra[4] = {make_uint4(qa[0].x, qa[0].y, qa[0].z, qa[0].w), make_uint4(qa[1].x, qa[1].y, qa[1].z, qa[1].w),
                             make_uint4(qa[2].x, qa[2].y, qa[2].z, qa[2].w), make_uint4(qa[3].x, qa[3].y, qa[3].z, qa[3].w)};
        quad_piece<2>(ra, cy, X, v, theta, mag4, refill_a, DUMP ? iqb : nullptr);
        reinterpret_cast<uint2 *>(dst)[0] = make_uint2(v[0], v[1]);
        reinterpret_cast<uint2 *>(dst)[1] = make_uint2(v[2], v[3]);
      }
      else
      {
        const uint4 ra[4] = {make_uint4(qa[0].x, qa[0].y, qa[0].z, qa[0].w), make_uint4(qa[1].x, qa[1].y, qa[1].z, qa[1].w),
                             make_uint4(qa[2].x, qa[2].y, qa[2].z, qa[2].w), make_uint4(qa[3].x, qa[3].y, qa[3].z, qa[3].w)};
        uint32_t mx[4];
        quad_front(ra, cy.fe, mx, refill_a);
        store_rails(mx, dst);
        if (DUMP)
        {
          iqb[0] = __builtin_amdgcn_perm(mx[1], mx[0], 0x06040200u) ^ 0x80808080u;   // (as quad_piece forms them)
          iqb[1] = __builtin_amdgcn_perm(mx[3], mx[2], 0x06040200u) ^ 0x80808080u;
        }
      }
      if (DUMP)
      {
        store_dump(iqb, u, 0);
      }
      magsum = mag4;
      uint32_t e0 = 0u, e1 = 0u, e2 = 0u, e3 = 0u;
      if (kWb && !kSplit)
      {
        e0 = (uint32_t)__builtin_amdgcn_readlane((int)f2u(theta[0]), 0);
        e1 = (uint32_t)__builtin_amdgcn_readlane((int)f2u(theta[1]), 0);
      }
      FLOW_MARK(4)
      FLOW_MARK(5)
      if (DUMP && !kDumpPlainWaits)
      {
        VM_WAIT(7, "+v"(qb[0]), "+v"(qb[1]), "+v"(qb[2]), "+v"(qb[3]));
      }
      else
      {
        VM_WAIT(5, "+v"(qb[0]), "+v"(qb[1]), "+v"(qb[2]), "+v"(qb[3]));
      }
      auto refill_b = [&](const uint32_t (&y1)[4][4]) { load_piece(qb, uoff_n, 1, un < n_units, y1); };
      if (kSplit)
      {
        const uint4 rb[4] = {make_uint4(qb[0].x, qb[0].y, qb[0].z, qb[0].w), make_uint4(qb[1].x, qb[1].y, qb[1].z, qb[1].w),
                             make_uint4(qb[2].x, qb[2].y, qb[2].z, qb[2].w), make_uint4(qb[3].x, qb[3].y, qb[3].z, qb[3].w)};
        uint32_t mx[4];
        quad_front(rb, cy.fe, mx, refill_b);
        store_pairs(mx, dst + 4 * kFStride2);
      }
      else if (kWb)
      {
        const uint4 rb[4] = {make_uint4(qb[0].x, qb[0].y, qb[0].z, qb[0].w), make_uint4(qb[1].x, qb[1].y, qb[1].z, qb[1].w),
                             make_uint4(qb[2].x, qb[2].y, qb[2].z, qb[2].w), make_uint4(qb[3].x, qb[3].y, qb[3].z, qb[3].w)};
        quad_piece<2>(rb, cy, X, v, theta, mag4, refill_b, DUMP ? iqb : nullptr);
        reinterpret_cast<uint2 *>(dst + 4 * kFStride)[0] = make_uint2(v[0], v[1]);
        reinterpret_cast<uint2 *>(dst + 4 * kFStride)[1] = make_uint2(v[2], v[3]);
      }
      else
      {
        const uint4 rb[4] = {make_uint4(qb[0].x, qb[0].y, qb[0].z, qb[0].w), make_uint4(qb[1].x, qb[1].y, qb[1].z, qb[1].w),
                             make_uint4(qb[2].x, qb[2].y, qb[2].z, qb[2].w), make_uint4(qb[3].x, qb[3].y, qb[3].z, qb[3].w)};
        uint32_t mx[4];
        quad_front(rb, cy.fe, mx, refill_b);
        store_rails(mx, dst + 4 * kFStride);
        if (DUMP)
        {
          iqb[0] = __builtin_amdgcn_perm(mx[1], mx[0], 0x06040200u) ^ 0x80808080u;
          iqb[1] = __builtin_amdgcn_perm(mx[3], mx[2], 0x06040200u) ^ 0x80808080u;
        }
      }
      if (DUMP)
      {
        store_dump(iqb, u, 1);
      }
      magsum += mag4;
      if (lastgen)
      {
        // the lane's eight thetas of this unit, from the ring words it has just stored (LDS executes a wave's operations in
        // order; here, behind both pieces, the front end's registers are dead: inside them the same code spilled):
        // tile 8 (u - first unit of the generation) + 4 half + lane / 16 of the generation, samples 4 (lane % 16) .. + 3 of it
        const int r0 = (64 * (n_gens - 1)) & (kFRing2 - 1);
        uint32_t *vs = ring + ((r0 >= 384) ? 0 : r0 + 64) * kFStride2;
#pragma unroll
        for (int half = 0; half < 2; half++)
        {
          const uint2 w = *reinterpret_cast<const uint2 *>(dst + half * 4 * kFStride2);
          const uint32_t x0 = w.x ^ 0x80808080u, x1 = w.y ^ 0x80808080u;
          const uint32_t a0 = abs4_s8(x0), a1 = abs4_s8(x1);
          const float t0 = theta_quad<0>(x0, a0, tquad), t1 = theta_quad<1>(x0, a0, tquad);
          const float t2 = theta_quad<0>(x1, a1, tquad), t3 = theta_quad<1>(x1, a1, tquad);
          const int tile = 8 * (u - 8 * (n_gens - 1)) + 4 * half + (lane >> 4);
          *reinterpret_cast<uint4 *>(vs + tile * kCoopStride + 4 * (lane & 15)) = make_uint4(f2u(t0), f2u(t1), f2u(t2), f2u(t3));
        }
      }
      if (kWb && !kSplit)
      {
        e2 = (uint32_t)__builtin_amdgcn_readlane((int)f2u(theta[2]), 63);
        e3 = (uint32_t)__builtin_amdgcn_readlane((int)f2u(theta[3]), 63);
      }
      FLOW_MARK(6)
      if (kWb && !kSplit && lane < 4)
      {
        edges[u & (kFEdges - 1)][lane] = (lane == 0) ? e0 : (lane == 1) ? e1 : (lane == 2) ? e2 : e3;
      }
      const bool counted = !GATED && u >= (hal >> 9);    // history in front of the run is not in any block's squelch sum (GATED: the batch launch's sums stand)
      int slot = 0;
      if (counted)
      {
        while (u >= bu0 + upb)
        {
          bu0 += upb;
          blk++;
        }
        slot = blk & 15;
        if (blk >= 16)
        {
          // The slot was block blk - 16's.  That block's units are 64 or more behind this one -- out of the ring or about
          // to be -- and the wave that counted its last unit finishes it at the top of its next iteration; only a wave that
          // was held up for many units' time at that very point is still to come (the stress build does that; blocks of
          // the shortest size, batches of more than 16).  Until round 5 nothing waited here: the late block's sum and
          // count took this unit's with them, the run's block count never became whole and the channel was replayed.
          const int ob = blk - 16;
          FlowSpin sp;
          while (((lds_ld(&ctl[kCtlBlk + ((ob >> 5) & 1)]) >> (ob & 31)) & 1u) == 0u && !sp.expired(P, ctl, fail_code, 15))
          {
            __builtin_amdgcn_s_sleep(2);
          }
          lds_order();
          if (fail_code != 0u)
          {
            break;
          }
        }
        atomicAdd(&magl[slot][lane], magsum);
      }
      // LDS executes a wave's operations in order: the flag and the block's unit count go out behind the data
      // without waiting for anything; the count's old value is looked at one unit later (pend_*)
      asm volatile("" ::: "memory");
      pend_nth = 0u;
      if (lane == 0)
      {
        lds_st(&uflag[u & (Lds::kEdges - 1)], (uint32_t)u + 1u);
        if (counted)
        {
          lds_add_async(pend_nth, &ctl[8 + slot], 1u);
        }
      }
      pend = counted;
      pend_slot = slot;
      pend_blk = blk;
      flow_hold_up(P, 7, u & 63);
      u = un;
      uoff = uoff_n;
      FLOW_MARK(7)
    }
    // The last prefetches (pointed outside the buffer) are still in flight, and the compiler does not know: their
    // registers must stay reserved until they have landed, or whatever reuses them is overwritten with zeros.
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(c16), "+v"(qa[0]), "+v"(qa[1]), "+v"(qa[2]), "+v"(qa[3]), "+v"(qb[0]), "+v"(qb[1]), "+v"(qb[2]), "+v"(qb[3])
                 :
                 : "memory");
    finish_block();
    FLOW_TIME_MAX(44)
    if (wave == SVC && !GATED)
    {
      // every stream wave is through its units by now or about to be: wait for the last blocks, then the
      // squelch inputs of the whole run go out
      const uint32_t nb = b_end - b_first;
      FlowSpin sp;
      while (lds_ld(&ctl[3]) != nb && !sp.expired(P, ctl, fail_code, 2))
      {
        __builtin_amdgcn_s_sleep(2);
      }
      lds_order();
      for (uint32_t i = lane; i < nb; i += 64)
      {
        const uint32_t w = blkout[i & 63];
        const uint32_t b = b_first + i;
        P.magnitude[(size_t)c * P.out_blocks + P.out_b0 + b] = w & 0x7fffffffu;
        P.present[(size_t)c * P.n_blocks + b] = (uint8_t)(w >> 31);
      }
    }
#ifdef HRFD_FLOW_PROBE
    if (P.dbg != nullptr && lane == 0)
    {
      for (int i = 0; i < 8; i++)
      {
        atomicAdd(&P.dbg[(size_t)blockIdx.x * kDbgSlots + 24 + i], probe[i]);
      }
    }
#endif
  }

  // --------------------------------------------------------------------------- service waves, FIR modes
  __device__ __forceinline__ void service_waves_fir()
  {
    setup_lds();                                         // (the stream waves call it behind their first requests)
    // =================================================================== service waves, FIR modes: rails -> PCM
    // Generations of 64 tiles as for WBFM, one tile (64 samples of both rails) per lane:
    //   a. the lane's tile through the FIRST decimator, straight out of the ring with the few samples of the tile in
    //      front -- AM / SSB: D(8,4) on both rails (AmDemodulator.cc:339-408, SsbDemodulator.cc:462-529); FM: the
    //      tuner D(32,4) on both rails, the table lookup on the low bytes and the "differentiator"
    //      theta[n-2] - theta[n-4] with its wrap and gain (FmDemodulator.cc:395-529);
    //   b. generations hand over in order (ctl[1]: the ring below is released; FM: the last four thetas go along);
    //   c. in order again (ctl[2]): the second decimator through the U and V rings -- FM: D(12,4), then D(40,2) and the
    //      PCM (FmDemodulator.cc:551-585); AM / SSB: D(12,4) per rail, and the generation is released: the third
    //      decimator D(16,2) needs nothing but V (its own and the last of the generation in front, written by then),
    //      so it runs beside the next generation's part c (the V ring holds four generations for that);
    //      at 8 kS/s the envelope (AM) or the negating delay line, the Hilbert transformer and I -/+ Q (SSB: the rails
    //      are published in order, gctl[3], nothing else waits);
    //   d. AM / SSB, in order once more (gctl[2]): the dc-removal recurrence y = (x - x1) - a1 y1
    //      (IirFilter.cc:161-176) SEQUENTIALLY over the generation's 128 samples on one lane, from the y the
    //      generation in front left: exact, nothing to verify -- then gain, (int16_t), PCM (AmDemodulator.cc:434-471,
    //      SsbDemodulator.cc:563-598).  Only this part is a chain through all generations: ~1.3 us of 3 per generation.
    __builtin_amdgcn_s_setprio(HRFD_FLOW_SVC_PRIO);
    const uint32_t pcm_off = (uint32_t)(hal >> 5);       // PCM samples that the history in front would yield
    uint32_t *pcm32 = reinterpret_cast<uint32_t *>(P.pcm + ((size_t)c * P.out_blocks + P.out_b0 + b_first) * (size_t)(n256 >> 5));
    const bool am = (MODE == 14) && cfg.mode == 1;
    const float gain8k = am ? cfg.gain_am : cfg.gain_ssb;
    float kg_fm = cfg.gain_fm / 15000.0f;                // K = (gain/15000)*32767 in float, that order (FmDemodulator.cc:487-490)
    kg_fm = kg_fm * 32767.0f;
    constexpr int kHalTiles = 8;                         // tiles of history in front of position 0
    auto grab_gen = [&]() -> int {
      uint32_t v = 0;
      if (lane == 0)
      {
        v = atomicAdd(&ctl[4], 1u);
      }
      return __builtin_amdgcn_readfirstlane((int)v);
    };
    // where PCM pair q / 2 of the stream goes (dword index).  GATED: the stream is a list of blocks
    auto pcm_at = [&](const uint32_t q) -> uint32_t {
      if (!GATED)
      {
        return q >> 1;
      }
      const uint32_t npcm = (uint32_t)n256 >> 5;
      const uint32_t k = q / npcm;
      return ((uint32_t)blist[k & 63u] * npcm + (q - k * npcm)) >> 1;
    };
    auto wait_for = [&](const uint32_t *word, const uint32_t want, const uint32_t where) {
      FlowSpin sp;
      while (lds_ld(word) != want && !sp.expired(P, ctl, fail_code, where))
      {
        __builtin_amdgcn_s_sleep(2);
      }
      lds_order();
    };
    for (int g = grab_gen(); g < n_gens; g = grab_gen())
    {
      const int t0 = 64 * g;
      const int ntl = min(64, n_tiles - t0);             // tiles of this generation (a multiple of 8)
      const int t = t0 + lane;
      const bool have = lane < ntl;
      // the generation's units, and the one in front (its last samples), are in the ring
      {
        const int ulo = max(8 * g - 1, 0), uhi = 8 * g + (ntl >> 3);
        const int uu = ulo + lane;
        FlowSpin sp;
        for (;;)
        {
          const bool ok = (uu >= uhi) || lds_ld(&uflag[uu & (Lds::kEdges - 1)]) == (uint32_t)uu + 1u;
          if (__all(ok) || sp.expired(P, ctl, fail_code, 3))
          {
            break;
          }
          __builtin_amdgcn_s_sleep(6);
        }
        lds_order();
      }
      if (fail_code != 0u)
      {
        break;
      }
      const uint32_t *tp = ring + ring_slot_n<Lds::kRingTiles>(t) * kFStride;
      const uint32_t *hp = ring + ring_slot_n<Lds::kRingTiles>(t > 0 ? t - 1 : 0) * kFStride;
      uint32_t ud[kRails][8];                            // the lane's 16 first-decimator outputs per rail, packed pairs
      float th[16];                                      // FM: theta of the lane's 16 samples at 64 kS/s
#pragma unroll
      for (int r = 0; r < kRails; r++)
      {
#pragma unroll
        for (int i = 0; i < 8; i++)
        {
          ud[r][i] = 0u;
        }
      }
#pragma unroll
      for (int i = 0; i < 16; i++)
      {
        th[i] = 0.0f;
      }
      // ---- a. the first decimator
      if (have)
      {
        if constexpr (MODE == 14)
        {
          constexpr int kC0 = q15_bias_init(Q_AM_D1);
#pragma unroll
          for (int r = 0; r < 2; r++)
          {
            uint32_t x[34];                              // pairs: samples -4 .. -1 of the tile, then the tile
            const uint2 h2 = (t > 0) ? *reinterpret_cast<const uint2 *>(hp + 32 * r + 30) : make_uint2(0x00800080u, 0x00800080u);
            x[0] = h2.x;
            x[1] = h2.y;
#pragma unroll
            for (int j = 0; j < 16; j++)
            {
              const uint2 w = *reinterpret_cast<const uint2 *>(tp + 32 * r + 2 * j);
              x[2 + 2 * j] = w.x;
              x[3 + 2 * j] = w.y;
            }
#pragma unroll
            for (int k = 0; k < 16; k++)
            {
              int acc = kC0;                             // y[k] over samples 4k-4 .. 4k+3 = pairs 2k .. 2k+3 of x
#pragma unroll
              for (int j = 0; j < 4; j++)
              {
                acc = dot2(x[2 * k + j], kRevAmD1.p[j], acc);
              }
              const uint32_t y = (uint32_t)q15_out(acc) & 0xffffu;
              ud[r][k >> 1] |= (k & 1) ? (y << 16) : y;
            }
          }
        }
        else
        {
          constexpr int kC0 = q15_bias_init(Q_FM_TUNER_D32);
          uint32_t lowb[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};   // low bytes of the tuner's outputs, biased by 128 (FmDemodulator.cc:495-496)
#pragma unroll
          for (int r = 0; r < 2; r++)
          {
            uint32_t x[46];                              // pairs: samples -28 .. -1 of the tile, then the tile
#pragma unroll
            for (int j = 0; j < 7; j++)
            {
              const uint2 w = (t > 0) ? *reinterpret_cast<const uint2 *>(hp + 32 * r + 18 + 2 * j) : make_uint2(0x00800080u, 0x00800080u);
              x[2 * j] = w.x;
              x[2 * j + 1] = w.y;
            }
#pragma unroll
            for (int j = 0; j < 16; j++)
            {
              const uint2 w = *reinterpret_cast<const uint2 *>(tp + 32 * r + 2 * j);
              x[14 + 2 * j] = w.x;
              x[15 + 2 * j] = w.y;
            }
#pragma unroll
            for (int k = 0; k < 16; k++)
            {
              int acc = kC0;                             // y[k] over samples 4k-28 .. 4k+3 = pairs 2k .. 2k+15 of x
#pragma unroll
              for (int j = 0; j < 16; j++)
              {
                acc = dot2(x[2 * k + j], kRevTuner.p[j], acc);
              }
              const uint32_t b = ((uint32_t)q15_out(acc) & 0xffu) ^ 0x80u;
              lowb[r][k >> 2] |= b << (8 * (k & 3));
            }
          }
#pragma unroll
          for (int k = 0; k < 16; k++)
          {
            const uint32_t ii = (lowb[0][k >> 2] >> (8 * (k & 3))) & 0xffu, qi = (lowb[1][k >> 2] >> (8 * (k & 3))) & 0xffu;
            th[k] = theta_tab((qi << 16) | ii, atcorr, att0);
          }
        }
      }
      lds_order();                                       // this wave's ring reads are done
      // ---- b. in order: the ring below this generation is released; FM: the thetas in front of the first lane
      wait_for(&ctl[1], (uint32_t)g, 5);
      if (fail_code != 0u)
      {
        break;
      }
      if constexpr (MODE == 2)
      {
        float thm[4];                                    // theta[-4 .. -1] of the lane's tile: the left lane's last four
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
          thm[j] = u2f(shr1(f2u(th[12 + j]), thfin[j]));
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
          const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)f2u(th[12 + j]), ntl - 1);
          if (lane == 0)
          {
            thfin[j] = v;
          }
        }
        // differentiator {0,0,1,0,-1,0,0} (FmDemodulator.cc:116-125: its -1/16 and 1/16 are integer divisions), wrap,
        // gain, (int16_t) narrowing (:567)
#pragma unroll
        for (int k = 0; k < 16; k++)
        {
          const float a = (k >= 2) ? th[k - 2] : thm[k + 2], b = (k >= 4) ? th[k - 4] : thm[k];
          const uint32_t y = (uint32_t)f2i16(kg_fm * wrap_pi(a - b)) & 0xffffu;
          ud[0][k >> 1] |= (k & 1) ? (y << 16) : y;
        }
      }
      asm volatile("" ::: "memory");
      if (lane == 0)
      {
        lds_st(&ctl[1], (uint32_t)g + 1u);
      }
      flow_hold_up(P, 1, g);
      // ---- c. in order: the stages behind the first decimator
      wait_for(&ctl[2], (uint32_t)g, 6);
      if (fail_code != 0u)
      {
        break;
      }
      if constexpr (MODE == 14)
      {
        // The V ring and SSB's 8 kS/s rings hold FOUR generations, and a generation reads the tail of the one in front
        // in its third decimator and its Hilbert transformer -- which run outside this ordered section, beside the
        // part c of the generations behind.  So before generation g writes over what g - 4 left, g - 3 must be through
        // them: through its part d (gctl[2], in order).  Normally it has been for a long time.  (Without this wait a
        // wave that is held up between its part c and its 8 kS/s part can be overtaken by the four generations behind
        // it: seen once, as one SSB channel of a mixed bank with wrong PCM in one launch of many.)
        FlowSpin sp;
        while (!(HRFD_ABLATE & 4096) && (int)lds_ld(&gctl[2]) < g - 2 && !sp.expired(P, ctl, fail_code, 7))   // (4096: TEST OF THE TEST ONLY)
        {
          __builtin_amdgcn_s_sleep(2);
        }
        lds_order();
        if (fail_code != 0u)
        {
          break;
        }
      }
      // (FM: U and V in front of position 0 are the carried pipelines -- the history tiles write neither)
      if (have && (MODE == 14 || t >= kHalTiles))
      {
#pragma unroll
        for (int r = 0; r < kRails; r++)
        {
          uint4 *up = reinterpret_cast<uint4 *>(uring + r * kFUDw + ((8 * t) & (kFUDw - 1)));
          up[0] = make_uint4(ud[r][0], ud[r][1], ud[r][2], ud[r][3]);
          up[1] = make_uint4(ud[r][4], ud[r][5], ud[r][6], ud[r][7]);
        }
      }
      // V[k] = D(12,4)(U), two per lane and pass
#pragma unroll
      for (int r = 0; r < kRails; r++)
      {
        for (int i = lane; i < 2 * ntl; i += 64)
        {
          const int k = 256 * g + 2 * i;                 // even
          const uint4 ua = *reinterpret_cast<const uint4 *>(uring + r * kFUDw + ((2 * k - 4) & (kFUDw - 1)));
          const uint4 ub = *reinterpret_cast<const uint4 *>(uring + r * kFUDw + ((2 * k) & (kFUDw - 1)));
          const uint32_t uu[8] = {ua.x, ua.y, ua.z, ua.w, ub.x, ub.y, ub.z, ub.w};   // U[4k-8 .. 4k+7]
          int acc0 = 1 << 14, acc1 = 1 << 14;
#pragma unroll
          for (int j = 0; j < 6; j++)
          {
            const uint32_t tap = (MODE == 14) ? kRevAmD2.p[j] : kRevD12.p[j];
            acc0 = dot2(uu[j], tap, acc0);
            acc1 = dot2(uu[j + 2], tap, acc1);
          }
          if (MODE == 14 || k >= 4 * kHalTiles)
          {
            vring[r * kVDw + ((k >> 1) & (kVDw - 1))] = ((uint32_t)q15_out(acc0) & 0xffffu) | ((uint32_t)q15_out(acc1) << 16);
          }
        }
      }
      if constexpr (MODE == 14)
      {
        // V of this generation is written: the next one may run its part c
        asm volatile("" ::: "memory");
        lds_order();
        if (lane == 0)
        {
          lds_st(&ctl[2], (uint32_t)g + 1u);
        }
        flow_hold_up(P, 2, g);
      }
      const int pp = 128 * g + 2 * lane;                 // the lane's two 8 kS/s samples (even index)
      if constexpr (MODE == 2)
      {
        // PCM[p] = D(40,2)(V), two per lane
        if (have)
        {
          int acc0 = 1 << 14, acc1 = 1 << 14;            // V[2pp-38 .. 2pp+3] = dwords pp-19 .. pp+1
          uint32_t prev = vring[(pp - 19) & (kFVDw - 1)];
#pragma unroll
          for (int j = 0; j < 20; j++)
          {
            const uint32_t next = vring[(pp - 18 + j) & (kFVDw - 1)];
            acc0 = dot2(prev, kRevD40.p[j], acc0);
            acc1 = dot2(next, kRevD40.p[j], acc1);
            prev = next;
          }
          if ((uint32_t)pp >= pcm_off)
          {
            pcm32[pcm_at((uint32_t)pp - pcm_off)] = ((uint32_t)q15_out(acc0) & 0xffffu) | ((uint32_t)q15_out(acc1) << 16);
          }
        }
      }
      else
      {
        // the third decimator D(16,2) on both rails: V[2pp-14 .. 2pp+3] = dwords pp-7 .. pp+1
        int o[2][2] = {{0, 0}, {0, 0}};
        if (have)
        {
#pragma unroll
          for (int r = 0; r < 2; r++)
          {
            uint32_t x[9];
#pragma unroll
            for (int j = 0; j < 9; j++)
            {
              x[j] = vring[r * kVDw + ((pp - 7 + j) & (kVDw - 1))];
            }
            int acc0 = 1 << 14, acc1 = 1 << 14;
#pragma unroll
            for (int j = 0; j < 8; j++)
            {
              acc0 = dot2(x[j], kRevAmD3.p[j], acc0);
              acc1 = dot2(x[j + 1], kRevAmD3.p[j], acc1);
            }
            o[r][0] = q15_out(acc0);
            o[r][1] = q15_out(acc1);
          }
        }
        // the recurrence's input of the lane's two samples.  AM: the envelope.  SSB: the rails at 8 kS/s go to their
        // rings (the samples in front of sample 0 are the carried ones; generations publish their rails in order,
        // gctl[3]: the Hilbert transformer reads 30 samples back, into the generation in front), then the negating
        // delay line (Q15 tap 1.0 narrows to -32768), the 31-tap Hilbert transformer, I -/+ Q
        float xv[2] = {0.0f, 0.0f};
        if (am)
        {
          // AmDemodulator::demodulateSignal (:447-461): int16 abs, compare, add with wrap
          auto env = [](int iv, int qv) -> int {
            const int im = (int)(short)abs(iv), qm = (int)(short)abs(qv);
            return (im > qm) ? (int)(short)(im + (qm >> 1)) : (int)(short)(qm + (im >> 1));
          };
          xv[0] = (float)env(o[0][0], o[1][0]);
          xv[1] = (float)env(o[0][1], o[1][1]);
        }
        else
        {
          if (have && pp >= 2 * kHalTiles)
          {
            reinterpret_cast<uint32_t *>(r8k[0])[(pp >> 1) & 255] = ((uint32_t)o[0][0] & 0xffffu) | ((uint32_t)o[0][1] << 16);
            reinterpret_cast<uint32_t *>(r8k[1])[(pp >> 1) & 255] = ((uint32_t)o[1][0] & 0xffffu) | ((uint32_t)o[1][1] << 16);
          }
          lds_order();
          wait_for(&gctl[3], (uint32_t)g, 4);
          if (fail_code != 0u)
          {
            break;
          }
          if (lane == 0)
          {
            lds_st(&gctl[3], (uint32_t)g + 1u);
          }
          flow_hold_up(P, 3, g);
          if (have)
          {
#pragma unroll
            for (int e = 0; e < 2; e++)
            {
              const int n = pp + e;
              const int idel = q15_out((1 << 14) + (-32768) * (int)r8k[0][(n - 15) & 511]);
              int acc = 1 << 14;
#pragma unroll
              for (int k = 0; k < N_SSB_HILBERT; k++)
              {
                acc += (int)Q_SSB_HILBERT[k] * (int)r8k[1][(n - k) & 511];
              }
              const int qh = q15_out(acc);
              xv[e] = (float)(cfg.lsb ? (idel - qh) : (idel + qh));
            }
          }
        }
        // ---- d. the recurrence, generation by generation
        wait_for(&gctl[2], (uint32_t)g, 4);
        if (fail_code != 0u)
        {
          break;
        }
        if (have)
        {
          *reinterpret_cast<float2 *>(&xs8k[2 * lane]) = make_float2(xv[0], xv[1]);
        }
        // the dc-removal recurrence over the generation's samples, one lane, groups of eight with the next group's
        // inputs in flight (the chain per step is the multiply and the subtract)
        if (lane == 0)
        {
          const int cnt = 2 * ntl;
          int n = max(0, 2 * kHalTiles - 128 * g);       // the history in front of sample 0 is not part of the stream
          float xp = rcar[0], y = rcar[1];
          float4 a0 = *reinterpret_cast<const float4 *>(&xs8k[n & 127]), a1 = *reinterpret_cast<const float4 *>(&xs8k[(n + 4) & 127]);
          for (; n < cnt; n += 8)
          {
            const float4 b0 = *reinterpret_cast<const float4 *>(&xs8k[(n + 8) & 127]), b1 = *reinterpret_cast<const float4 *>(&xs8k[(n + 12) & 127]);
            const float xin[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            float yo[8];
#pragma unroll
            for (int j = 0; j < 8; j++)
            {
              const float v = xin[j] - xp;
              xp = xin[j];
              const float r = DCREM_A1 * y;
              y = v - r;
              yo[j] = y;
            }
            *reinterpret_cast<float4 *>(&ys8k[n]) = make_float4(yo[0], yo[1], yo[2], yo[3]);
            *reinterpret_cast<float4 *>(&ys8k[n + 4]) = make_float4(yo[4], yo[5], yo[6], yo[7]);
            a0 = b0;
            a1 = b1;
          }
          rcar[0] = xp;
          rcar[1] = y;
        }
        if (have && (uint32_t)pp >= pcm_off)
        {
          const float2 yy = *reinterpret_cast<const float2 *>(&ys8k[2 * lane]);
          pcm32[pcm_at((uint32_t)pp - pcm_off)] = ((uint32_t)f2i16(gain8k * yy.x) & 0xffffu) | ((uint32_t)f2i16(gain8k * yy.y) << 16);
        }
      }
      // ---- the carried state for the next call (pending: the finisher commits it)
      if (g + 1 == n_gens)
      {
        // the demodulator's input tail: the last samples of the stream as offset-binary bytes i, q (the ring's pairs)
        constexpr int kTailPairs = (MODE == 2 ? kFmTail : kAmTail) / 2;
        uint32_t *tail = reinterpret_cast<uint32_t *>((MODE == 2) ? so->fm_tail : am ? so->am_tail : so->ssb_tail);
        for (int m = lane; m < kTailPairs; m += 64)
        {
          const int pr = 32 * n_tiles - kTailPairs + m;  // pair index in the stream
          const uint32_t *sp = ring + ring_slot_n<Lds::kRingTiles>(pr >> 5) * kFStride + (pr & 31);
          const uint32_t ip = sp[0], qp = sp[32];
          tail[m] = (ip & 0xffu) | ((qp & 0xffu) << 8) | (ip & 0x00ff0000u) | ((qp & 0x00ff0000u) << 8);
        }
        if constexpr (MODE == 2)
        {
          if (lane < 4)
          {
            reinterpret_cast<uint32_t *>(so->fm_u)[lane] = uring[(8 * n_tiles - 4 + lane) & (kFUDw - 1)];
          }
          if (lane < 19)
          {
            reinterpret_cast<uint32_t *>(so->fm_v)[lane] = vring[(2 * n_tiles - 19 + lane) & (kFVDw - 1)];
          }
        }
        else if (am)
        {
          if (lane == 0)
          {
            so->am_x1 = rcar[0];
            so->am_y1 = rcar[1];
          }
        }
        else
        {
          if (lane == 0)
          {
            so->ssb_x1 = rcar[0];
            so->ssb_y1 = rcar[1];
          }
          if (lane < 16)
          {
            reinterpret_cast<uint32_t *>(so->ssb_i)[lane] = reinterpret_cast<const uint32_t *>(r8k[0])[(n_tiles - 16 + lane) & 255];
            reinterpret_cast<uint32_t *>(so->ssb_q)[lane] = reinterpret_cast<const uint32_t *>(r8k[1])[(n_tiles - 16 + lane) & 255];
          }
        }
      }
      asm volatile("" ::: "memory");
      lds_order();
      if (lane == 0)
      {
        lds_st((MODE == 14) ? &gctl[2] : &ctl[2], (uint32_t)g + 1u);
      }
    }
  }

  // -------------------------------------------------------------------------------- service waves, WBFM
  __device__ __forceinline__ void service_waves_wbfm()
  {
    setup_lds();                                         // (the stream waves call it behind their first requests)
    // =================================================================== service waves: v -> PCM
    __builtin_amdgcn_s_setprio(HRFD_FLOW_SVC_PRIO);      // long dependent chains, few issue slots
    const float a1 = DEEMPH_A1;
    const int fa_t = first ? 0 : (wt + M);               // first tile that can be started properly
    const uint32_t pcm_off = (uint32_t)(hal >> 5);       // PCM samples that the history in front would yield
    uint32_t *pcm32 = reinterpret_cast<uint32_t *>(P.pcm + ((size_t)c * P.out_blocks + P.out_b0 + b_first) * (size_t)(n256 >> 5));
    uint32_t repairs = 0;
    unsigned long long sprobe[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, sprev = __builtin_readcyclecounter();
    (void)sprobe; (void)sprev;
    // generations are taken in order by whichever service wave is free (a fixed rotation makes generation g wait
    // behind g - SVC on a wave that is late while the others idle)
    auto grab_gen = [&]() -> int {
      uint32_t v = 0;
      if (lane == 0)
      {
        v = atomicAdd(&ctl[4], 1u);
      }
      return __builtin_amdgcn_readfirstlane((int)v);
    };
    for (int g = grab_gen(); g < n_gens && !(HRFD_ABLATE & 1024); g = grab_gen())
    {
      SVC_MARK(0)
      const int t0 = 64 * g;
      const int ntl = min(64, n_tiles - t0);             // tiles of this generation (a multiple of 8)
      const int t = t0 + lane;
      const bool have = lane < ntl;
      // 1. the generation's units, and the one in front (its last two thetas), are in the ring
      {
        const int ulo = max(8 * g - 1, 0), uhi = 8 * g + (ntl >> 3);
        const int uu = ulo + lane;
        const unsigned long long tw0 = __builtin_readcyclecounter();
        FlowSpin sp;
        for (;;)
        {
          const bool ok = (uu >= uhi) || lds_ld(&uflag[uu & (kFEdges - 1)]) == (uint32_t)uu + 1u;
          if (__all(ok) || sp.expired(P, ctl, fail_code, 3))
          {
            break;
          }
          __builtin_amdgcn_s_sleep(6);
        }
        waited += __builtin_readcyclecounter() - tw0;
        lds_order();
      }
      if (fail_code != 0u)
      {
        break;                                           // the workgroup is aborting (FlowSpin): the channel will be replayed
      }
      SVC_MARK(1)
      // 2. the two provisional samples at the start of every unit (they need theta of the two samples in
      //    front of it, which another wave produced); one lane per unit
      if (lane < (ntl >> 3))
      {
        const int uu = 8 * g + lane;
        if (uu > 0)
        {
          const uint32_t *ep = edges[(uu - 1) & (kFEdges - 1)], *ec = edges[uu & (kFEdges - 1)];
          const float tm2 = u2f(ep[2]), tm1 = u2f(ep[3]);
          const float th0 = u2f(ec[0]), th1 = u2f(ec[1]);
          const float pm1 = numerator_p(tm1, tm2, kgain);
          const float p0 = numerator_p(th0, tm1, kgain);
          const float p1 = numerator_p(th1, th0, kgain);
          uint32_t *vp = ring + ring_slot(8 * uu) * kFStride;
          vp[0] = f2u(p0 + pm1);
          vp[1] = f2u(p1 + p0);
        }
      }
      const uint32_t *tp = ring + ring_slot(t) * kFStride;
      SVC_MARK(2)
      // 3. geometric partial sum of v over the own tile: P = sum_k c^k v[63 - k], c = -a1 (approximate on purpose)
      if (have && M > 0)
      {
        // (the even and the odd samples as the two halves of one packed fma: a lone wave is bound by its instruction count)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const float cc = -a1, c2 = cc * cc;
        const f32x2 c22 = {c2, c2};
        const f32x2 *p2 = reinterpret_cast<const f32x2 *>(tp);
        f32x2 pab = {0.0f, 0.0f};
#pragma unroll 8
        for (int j = 0; j < kFT / 2; j++)
        {
          pab = __builtin_elementwise_fma(pab, c22, p2[j]);
        }
        float p = __builtin_fmaf(pab.x, cc, pab.y);
        if (first && t == 0)
        {
          p += deemph_pow(kFT) * st->wb_y;               // the stream's past, as seen from the end of tile 0
        }
        parr[t & (kFPRing - 1)] = p;
      }
      lds_order();
      if (lane == 0)
      {
        lds_st(&pflag[g & 7], (uint32_t)g + 1u);
      }
      flow_hold_up(P, 4, g);
      SVC_MARK(3)
      if (g > 0 && M > 0)
      {
        const unsigned long long tw0 = __builtin_readcyclecounter();
        FlowSpin sp;
        while (lds_ld(&pflag[(g - 1) & 7]) != (uint32_t)g && !sp.expired(P, ctl, fail_code, 4))
        {
          __builtin_amdgcn_s_sleep(4);
        }
        waited += __builtin_readcyclecounter() - tw0;
        lds_order();
        if (fail_code != 0u)
        {
          break;
        }
      }
      SVC_MARK(4)
      // 4. seed, warm-up, tile
      const bool active = have && t >= fa_t;
      FlowTile o;
      o.y = 0.0f;
      o.sfirst0 = o.sfirst1 = o.slast0 = o.slast1 = 0u;
#pragma unroll
      for (int i = 0; i < 8; i++)
      {
        o.ud[i] = 0u;
      }
      float y_spec = 0.0f;
      if (active)
      {
        const int ws = t - wt;                           // the warm-up begins with this tile
        float y = 0.0f;
        if (first && ws <= 0)
        {
          y = st->wb_y;                                  // exact: the stream starts here
        }
        else if (M > 0)
        {
          // y at the end of tile ws - 1: sum_m (c^64)^(m-1) P[ws - m], oldest first
          float acc = 0.0f;
          for (int m = M; m >= 1; m--)
          {
            const int idx = ws - m;
            const float pv = (idx >= 0) ? parr[idx & (kFPRing - 1)] : 0.0f;
            acc = __builtin_fmaf(acc, P.flow_seed_ct, pv);
          }
          y = acc;
        }
        for (int k = wt; k >= 1; k--)
        {
          const int tw = t - k;
          if (tw >= 0)
          {
            y = flow_warm_tile(ring + ring_slot(tw) * kFStride, y);
          }
        }
        y_spec = y;
        if (small_y)
        {
          flow_tile_u<false>(tp, y, o);
        }
        else
        {
          flow_tile_u<true>(tp, y, o);
        }
      }
      SVC_MARK(5)
      // 5. generations are verified in order ...
      {
        const unsigned long long tw0 = __builtin_readcyclecounter();
        FlowSpin sp;
        while (lds_ld(&ctl[1]) != (uint32_t)g && !sp.expired(P, ctl, fail_code, 5))
        {
          __builtin_amdgcn_s_sleep(2);
        }
        waited += __builtin_readcyclecounter() - tw0;
        lds_order();
      }
      if (fail_code != 0u)
      {
        break;
      }
      SVC_MARK(6)
      const uint32_t left_y = wfin[0], left_s0 = wfin[1], left_s1 = wfin[2];
      // 6. every lane but the first runnable one checks its speculated start against its left neighbour's end;
      //    a tile that has not merged is re-run from the true value, ascending (rare)
      {
        const float y_left = u2f(shr1(f2u(o.y), left_y));
        const bool bad = active && t > fa_t && !same_trajectory(y_left, y_spec);
        unsigned long long bm = __ballot(bad);
        while (bm != 0ull)
        {
          const int l = __ffsll((long long)bm) - 1;      // wave-uniform
          bm &= ~(1ull << l);
          repairs++;
          const float y_true = u2f(shr1(f2u(o.y), left_y));
          if (lane == l)
          {
            if (small_y)
            {
              flow_tile_u<false>(tp, y_true, o);
            }
            else
            {
              flow_tile_u<true>(tp, y_true, o);
            }
          }
          // the right neighbour's speculation must now match the corrected final y
          const float y_new_left = u2f(shr1(f2u(o.y), left_y));
          const bool bad2 = (lane == l + 1) && active && !same_trajectory(y_new_left, y_spec);
          bm |= __ballot(bad2);
        }
      }
      // ... and hand their last lane's end to the next one at once: from here on nobody reads this generation's v
      // (but for the next one's warm-up over its last tiles, which the ring check of the stream waves allows for)
      const uint32_t fy = (uint32_t)__builtin_amdgcn_readlane((int)f2u(o.y), ntl - 1);
      const uint32_t fs0 = (uint32_t)__builtin_amdgcn_readlane((int)o.slast0, ntl - 1);
      const uint32_t fs1 = (uint32_t)__builtin_amdgcn_readlane((int)o.slast1, ntl - 1);
      if (lane == 0)
      {
        wfin[0] = fy;
        wfin[1] = fs0;
        wfin[2] = fs1;
      }
      asm volatile("" ::: "memory");
      if (lane == 0)
      {
        lds_st(&ctl[1], (uint32_t)g + 1u);
      }
      flow_hold_up(P, 5, g);
      // the integer stages follow in a chain of their own
      {
        const unsigned long long tw0 = __builtin_readcyclecounter();
        FlowSpin sp;
        while (lds_ld(&ctl[2]) != (uint32_t)g && !sp.expired(P, ctl, fail_code, 6))
        {
          __builtin_amdgcn_s_sleep(2);
        }
        waited += __builtin_readcyclecounter() - tw0;
        lds_order();
      }
      if (fail_code != 0u)
      {
        break;
      }
      // 7. U[0] of every tile: S[-4 .. -1] are the LEFT lane's last four samples (its final ones)
      {
        const uint32_t ls0 = shr1(o.slast0, left_s0), ls1 = shr1(o.slast1, left_s1);
        int acc = 1 << 14;
        acc = dot2(ls0, kRevWbD1.p[0], acc);
        acc = dot2(ls1, kRevWbD1.p[1], acc);
        acc = dot2(o.sfirst0, kRevWbD1.p[2], acc);
        acc = dot2(o.sfirst1, kRevWbD1.p[3], acc);
        o.ud[0] = (o.ud[0] & 0xffff0000u) | ((uint32_t)q15_out(acc) & 0xffffu);
      }
      if (have)
      {
        uint4 *up = reinterpret_cast<uint4 *>(uring + ((8 * t) & (kFUDw - 1)));
        up[0] = make_uint4(o.ud[0], o.ud[1], o.ud[2], o.ud[3]);
        up[1] = make_uint4(o.ud[4], o.ud[5], o.ud[6], o.ud[7]);
      }
      SVC_MARK(7)
      // 8. V[k] = D(12,4)(U), two per lane and pass (WbFmDemodulator.cc:478-486)
      for (int i = lane; i < 2 * ntl; i += 64)
      {
        const int k = 256 * g + 2 * i;                   // even
        const uint4 ua = *reinterpret_cast<const uint4 *>(uring + ((2 * k - 4) & (kFUDw - 1)));
        const uint4 ub = *reinterpret_cast<const uint4 *>(uring + ((2 * k) & (kFUDw - 1)));
        const uint32_t uu[8] = {ua.x, ua.y, ua.z, ua.w, ub.x, ub.y, ub.z, ub.w};   // U[4k-8 .. 4k+7]
        int acc0 = 1 << 14, acc1 = 1 << 14;
#pragma unroll
        for (int j = 0; j < 6; j++)
        {
          acc0 = dot2(uu[j], kRevD12.p[j], acc0);
          acc1 = dot2(uu[j + 2], kRevD12.p[j], acc1);
        }
        vring[(k >> 1) & (kFVDw - 1)] = ((uint32_t)q15_out(acc0) & 0xffffu) | ((uint32_t)q15_out(acc1) << 16);
      }
      // 9. PCM[p] = D(40,2)(V), two per lane (WbFmDemodulator.cc:488-496)
      if (have)
      {
        const int pp = 128 * g + 2 * lane;               // even; V[2pp-38 .. 2pp+3] = dwords pp-19 .. pp+1
        int acc0 = 1 << 14, acc1 = 1 << 14;
        uint32_t prev = vring[(pp - 19) & (kFVDw - 1)];
#pragma unroll
        for (int j = 0; j < 20; j++)
        {
          const uint32_t next = vring[(pp - 18 + j) & (kFVDw - 1)];
          acc0 = dot2(prev, kRevD40.p[j], acc0);
          acc1 = dot2(next, kRevD40.p[j], acc1);
          prev = next;
        }
        if (GATED)
        {
          // the stream is a list of blocks: PCM pair pp / 2 of the stream is pair (pp mod npcm) / 2 of block blist[pp / npcm]
          const uint32_t npcm = (uint32_t)n256 >> 5;
          const uint32_t k = (uint32_t)pp / npcm;
          pcm32[((uint32_t)blist[k & 63u] * npcm + ((uint32_t)pp - k * npcm)) >> 1] =
              ((uint32_t)q15_out(acc0) & 0xffffu) | ((uint32_t)q15_out(acc1) << 16);
        }
        else if ((uint32_t)pp >= pcm_off)
        {
          pcm32[((uint32_t)pp - pcm_off) >> 1] = ((uint32_t)q15_out(acc0) & 0xffffu) | ((uint32_t)q15_out(acc1) << 16);
        }
      }
      SVC_MARK(8)
      // 10. cross-block check values: y at block-relative position -705 = the end of the tile [-768, -704)
      //     (GATED: one exact stream from the committed state, nothing to check)
      if (have && !GATED)
      {
        const int x = 64 * t + 768 - hal;                // = (number of blocks completed) * n256 when this is such a tile
        if (x >= 0)
        {
          const int q = x / n256;
          if (q * n256 == x)
          {
            const uint32_t b = b_first + (uint32_t)q;    // the block this value stands in front of
            if (q == 0)
            {
              P.chk_spec[(size_t)c * P.n_blocks + b] = o.y;   // speculated by this run (b_first > 0)
            }
            else
            {
              P.chk_pub[(size_t)c * P.n_blocks + b - 1] = o.y;
              if (b < b_end)
              {
                P.chk_spec[(size_t)c * P.n_blocks + b] = o.y;
                if (local)
                {
                  lds_st(&finl[b & 63u], f2u(o.y));
                }
              }
            }
          }
        }
      }
      // 11. hand over to the next generation
      {
        if (g + 1 == n_gens && (GATED || b_end == P.n_blocks))
        {
          // the carried state for the next call (pending: k_rx_commit copies it when the launch verified clean)
          const uint32_t *el = edges[(n_units - 1) & (kFEdges - 1)];
          if (lane == 0)
          {
            so->wb_y = u2f(fy);
            so->wb_theta = u2f(el[3]);
            so->wb_p = numerator_p(u2f(el[3]), u2f(el[2]), kgain);
            reinterpret_cast<uint32_t *>(so->wb_s)[0] = fs0;
            reinterpret_cast<uint32_t *>(so->wb_s)[1] = fs1;
          }
          if (lane < 4)
          {
            reinterpret_cast<uint32_t *>(so->wb_u)[lane] = uring[(8 * n_tiles - 4 + lane) & (kFUDw - 1)];
          }
          if (lane < 19)
          {
            reinterpret_cast<uint32_t *>(so->wb_v)[lane] = vring[(2 * n_tiles - 19 + lane) & (kFVDw - 1)];
          }
          if (local)
          {
            // the same section in ChanState's order, dwords from wb_theta on: theta, p, y, pad, s (2), u (4), v (20)
            static_assert(offsetof(ChanState, wb_s) - offsetof(ChanState, wb_theta) == 16 && offsetof(ChanState, wb_u) - offsetof(ChanState, wb_theta) == 24 &&
                          offsetof(ChanState, wb_v) - offsetof(ChanState, wb_theta) == 40 && offsetof(ChanState, fm_tail) - offsetof(ChanState, wb_theta) == 120, "finl");
            if (lane == 0)
            {
              lds_st(&finl[128], el[3]);
              lds_st(&finl[129], f2u(numerator_p(u2f(el[3]), u2f(el[2]), kgain)));
              lds_st(&finl[130], fy);
              lds_st(&finl[131], 0u);
              lds_st(&finl[132], fs0);
              lds_st(&finl[133], fs1);
            }
            if (lane < 4)
            {
              lds_st(&finl[134 + lane], uring[(8 * n_tiles - 4 + lane) & (kFUDw - 1)]);
            }
            if (lane < 20)
            {
              lds_st(&finl[138 + lane], lane < 19 ? vring[(2 * n_tiles - 19 + lane) & (kFVDw - 1)] : 0u);
            }
          }
        }
        asm volatile("" ::: "memory");
        if (lane == 0)
        {
          lds_st(&ctl[2], (uint32_t)g + 1u);
        }
      }
      flow_hold_up(P, 6, g);
      SVC_MARK(9)
    }
    FLOW_TIME_MAX(45)
#ifdef HRFD_FLOW_PROBE
    if (P.dbg != nullptr && lane == 0)
    {
      for (int i = 0; i < 10; i++)
      {
        atomicAdd(&P.dbg[(size_t)blockIdx.x * kDbgSlots + 32 + i], sprobe[i]);
      }
    }
#endif
    if (repairs != 0u && lane == 0)
    {
      atomicAdd(&P.counters[kCntRepair], repairs);
      atomicAdd(&P.sticky[kCntTotRepair], repairs);
    }
  }

  // ------------------------------------------------------ theta / wrap / numerator of 8 NG samples (re-split)
  // w: NG groups of four ring words (bytes i0 q0 i1 q1, offset binary); thp / pp: theta and b0*x of the sample in front,
  // updated to the last sample's; v: the FIR half of the de-emphasis filter (WbFmDemodulator.cc:404-430,
  // IirFilter.cc:161-176: v = b0 x + b1 x[n-1], b1 == b0).  Eight table lookups are in flight before the first is used:
  // behind each other every pair exposes the LDS latency -- a quarter of this stretch when the wave is alone on its SIMD.
  template <int NG>
  __device__ __forceinline__ void theta_to_v(const uint32_t (&w)[4 * NG], float &thp, float &pp, float (&v)[8 * NG])
  {
#pragma unroll
    for (int j = 0; j < NG; j++)
    {
      uint32_t xs[4], tw[8];
#pragma unroll
      for (int k = 0; k < 4; k++)
      {
        xs[k] = w[4 * j + k] ^ 0x80808080u;
        const uint32_t a = abs4_s8(xs[k]);
        tw[2 * k] = tquad[theta_quad_index<0>(a)];
        tw[2 * k + 1] = tquad[theta_quad_index<1>(a)];
      }
      asm volatile("" ::: "memory");                     // (the compiler may not sink the reads back to their uses)
#pragma unroll
      for (int k = 0; k < 4; k++)
      {
        const float th0 = theta_quad_word<0>(xs[k], tw[2 * k]), th1 = theta_quad_word<1>(xs[k], tw[2 * k + 1]);
        const float p0 = numerator_p<true>(th0, thp, kgain);
        const float p1 = numerator_p<true>(th1, th0, kgain);
        v[8 * j + 2 * k] = p0 + pp;
        v[8 * j + 2 * k + 1] = p1 + p0;
        thp = th1;
        pp = p1;
      }
    }
  }
  // theta and b0*x of the two samples in the ring word in front of a stretch (tile 0 of the stream, quarter 0: the
  // carried ones when the stream continues the previous call; a run that re-derives its history starts from zeros --
  // on silent input that IS the truth: a transient there would never die away bit for bit and fail the cross-run check)
  __device__ __forceinline__ void theta_in_front(const uint32_t prevw, const bool stream_start, const float theta_in, const float p_in,
                                                 float &thp, float &pp)
  {
    const uint32_t x = prevw ^ 0x80808080u, a = abs4_s8(x);
    const float tm2 = theta_quad<0>(x, a, tquad), tm1 = theta_quad<1>(x, a, tquad);
    thp = tm1;
    pp = numerator_p<true>(tm1, tm2, kgain);
    if (stream_start)
    {
      thp = first ? theta_in : 0.0f;
      pp = first ? p_in : 0.0f;
    }
  }
  // The last generation's quarters (kCtlCoop*): claimed one at a time by whoever is there -- the generation's own wave
  // and the service waves that have run out of generations -- until all four are taken
  // FROM_END: the helpers take the quarters from the other end, so that they and the generation's wave meet in the middle.
  // (A loop that runs until a claim counter says "none left" never came back from the GPU, with atomicAdd() and with the
  // inline-assembly add alike, while the same body under a fixed four-turn loop did -- bring-up builds
  // -DHRFD_FLOW_COOP=2 / 3.  Hence four fixed turns and one claim BIT per quarter: the loop's shape does not depend on
  // what the other waves do, only whether a turn's body runs.)
  template <bool FROM_END>
  __device__ __forceinline__ void coop_quarters(const int g, const float theta_in, const float p_in)
  {
    const int t0 = 64 * g;
    const int ntl = min(64, n_tiles - t0);
    const int t = t0 + lane;
    const int r0 = t0 & (kFRing2 - 1);
    uint32_t *vs = ring + ((r0 >= 384) ? 0 : r0 + 64) * kFStride2;   // rows that nobody reads or writes any more
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
      const int q = FROM_END ? 3 - k : k;
      uint32_t old = 0;
      if (lane == 0)
      {
        lds_or_async(old, &ctl[kCtlCoopNext], 1u << q);
      }
      lds_landed(old);
      const bool taken = (((uint32_t)__builtin_amdgcn_readfirstlane((int)old) >> q) & 1u) != 0u;
      if (!taken)
      {
        if (lane < ntl)
        {
          const uint32_t *tp = ring + (t & (kFRing2 - 1)) * kFStride2;
          const uint4 a = reinterpret_cast<const uint4 *>(tp)[2 * q], b = reinterpret_cast<const uint4 *>(tp)[2 * q + 1];
          const uint32_t *pp_ = (q != 0) ? tp + 8 * q - 1
                                : (lane == 0) ? &lastdw[(g - 1) & 7] : ring + ((t - 1) & (kFRing2 - 1)) * kFStride2 + 31;
          const uint32_t prevw = *pp_;
          float thp, pp;
          theta_in_front(prevw, t == 0 && q == 0, theta_in, p_in, thp, pp);
          const uint32_t w8[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
          float v16[16];
          theta_to_v<2>(w8, thp, pp, v16);
          uint4 *dq = reinterpret_cast<uint4 *>(vs + lane * kCoopStride + 16 * q);
#pragma unroll
          for (int i = 0; i < 4; i++)
          {
            dq[i] = make_uint4(f2u(v16[4 * i]), f2u(v16[4 * i + 1]), f2u(v16[4 * i + 2]), f2u(v16[4 * i + 3]));
          }
          if (q == 3)
          {
            vs[64 * kCoopStride + lane] = f2u(thp);
            vs[64 * kCoopStride + 64 + lane] = f2u(pp);
          }
        }
        asm volatile("" ::: "memory");
        if (lane == 0)
        {
          atomicAdd(&ctl[kCtlCoopDone], 1u);             // (LDS runs a wave's operations in order: behind its stores)
        }
      }
    }
  }

  // --------------------------------------------------------------- service waves, WBFM, re-split (round 5)
  __device__ __forceinline__ void service_waves_wbfm2()
  {
    setup_lds();                                         // (the stream waves call it behind their first requests)
    // =================================================================== service waves: (q, i) pairs -> PCM
    // A generation is 64 tiles, one per lane, as in service_waves_wbfm -- but the lane makes its tile's v ITSELF, from the
    // ring's (q, i) pairs (theta_quad, wrap, gain, the FIR half of the de-emphasis filter: WbFmDemodulator.cc:404-430,
    // IirFilter.cc:161-176), and keeps the 64 floats in REGISTERS through everything that reads them:
    //   the geometric partial sum; `wt` WARM-UP PASSES over the lane's OWN tile -- pass k starts from the value the LEFT
    //   lane's pass k - 1 ended with (one DPP hop; lane 0 takes the last lane of the generation in front from wcar[]),
    //   pass 0 from the seed at the end of the tile in front: after wt hops lane l starts its tile from exactly the value
    //   the round-4 kernel computed by running tiles l - wt .. l - 1 itself out of the ring -- the same operations on the
    //   same operands, so the same bits, the same verification and the same (rare) repairs -- and then the tile proper.
    // The ring's rows are read ONCE, in generation order, and handed back to the stream waves at once (kCtlRel).
    // (the per-generation records -- lastdw, wcar / wflag, parr, pflag -- have eight slots: at most SVC generations are
    //  in flight, consecutive ones, because a generation completes only behind the one in front: verification order)
    static_assert(SVC <= 7, "per-generation records of the re-split service waves");
    __builtin_amdgcn_s_setprio(HRFD_FLOW_SVC_PRIO_WB);
    const float a1 = DEEMPH_A1;
    const int fa_t = first ? 0 : (wt + M);               // first tile that can be started properly
    const uint32_t pcm_off = (uint32_t)(hal >> 5);       // PCM samples that the history in front would yield
    uint32_t *pcm32 = reinterpret_cast<uint32_t *>(P.pcm + ((size_t)c * P.out_blocks + P.out_b0 + b_first) * (size_t)(n256 >> 5));
    uint32_t repairs = 0;
    unsigned long long sprobe[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, sprev = __builtin_readcyclecounter();
    (void)sprobe; (void)sprev;
    // the carried state of a stream that continues the previous call (requested before the table: one round trip)
    // (wave-uniform, and said so: three scalar registers instead of three vector ones held for the whole stream)
    const float y_in = u2f((uint32_t)__builtin_amdgcn_readfirstlane((int)f2u(st->wb_y)));
    const float theta_in = u2f((uint32_t)__builtin_amdgcn_readfirstlane((int)f2u(st->wb_theta)));
    const float p_in = u2f((uint32_t)__builtin_amdgcn_readfirstlane((int)f2u(st->wb_p)));
    // The table comes into LDS through the service waves while the stream waves already run (they do not need it):
    // 66 KB per workgroup that the round-4 kernel loaded in front of everything.
    {
      // (every load is requested before the first is stored: ONE memory round trip -- a copy loop costs one per turn,
      //  11 of them at 1.5 us each while the other CUs saturate the memory system: measured, 19 us)
      const uint4 *src = reinterpret_cast<const uint4 *>(P.at_quad);
      uint4 *dstq = reinterpret_cast<uint4 *>(tquad);
      constexpr int kTurns = (kQuadDwords / 4 + 64 * SVC - 1) / (64 * SVC);
      uint4 tq[kTurns];
#pragma unroll
      for (int k = 0; k < kTurns; k++)
      {
        const int i = tid + 64 * SVC * k;
        tq[k] = (i < kQuadDwords / 4) ? src[i] : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int k = 0; k < kTurns; k++)
      {
        const int i = tid + 64 * SVC * k;
        if (i < kQuadDwords / 4)
        {
          dstq[i] = tq[k];
        }
      }
      if (lane == 0)
      {
        atomicAdd(&ctl[kCtlTab], 1u);                    // (LDS runs a wave's operations in order: behind its copies)
      }
      FlowSpin sp;
      while (lds_ld(&ctl[kCtlTab]) != (uint32_t)SVC && !sp.expired(P, ctl, fail_code, 9))
      {
        __builtin_amdgcn_s_sleep(2);
      }
      lds_order();
    }
    FLOW_TIME_SET(43)
    auto grab_gen = [&]() -> int {
      uint32_t v = 0;
      if (lane == 0)
      {
        v = atomicAdd(&ctl[4], 1u);
      }
      return __builtin_amdgcn_readfirstlane((int)v);
    };
    float th_last = 0.0f, p_last = 0.0f;                 // theta and b0*x of the lane's last sample (the carried state at the end)
    for (int g = grab_gen(); g < n_gens && fail_code == 0u && !(HRFD_ABLATE & 1024); g = grab_gen())
    {
      SVC_MARK(0)
      const int t0 = 64 * g;
      const int ntl = min(64, n_tiles - t0);             // tiles of this generation (a multiple of 8)
      const int t = t0 + lane;
      const bool have = lane < ntl;
      // 1. the generation's units are in the ring ...
      {
        const int ulo = 8 * g, uhi = 8 * g + (ntl >> 3);
        const int uu = ulo + lane;
        const unsigned long long tw0 = __builtin_readcyclecounter();
        FlowSpin sp;
        for (;;)
        {
          const bool ok = (uu >= uhi) || lds_ld(&uflag[uu & (kFEdges2 - 1)]) == (uint32_t)uu + 1u;
          if (__all(ok) || sp.expired(P, ctl, fail_code, 3))
          {
            break;
          }
          __builtin_amdgcn_s_sleep(6);
        }
        // ... and the generation in front has read its rows (they are read in order: lastdw[] of it is there)
        while (lds_ld(&ctl[kCtlRel]) != (uint32_t)g && !sp.expired(P, ctl, fail_code, 10))
        {
          __builtin_amdgcn_s_sleep(2);
        }
        waited += __builtin_readcyclecounter() - tw0;
        lds_order();
      }
      if (fail_code != 0u)
      {
        break;                                           // the workgroup is aborting (FlowSpin): the channel will be replayed
      }
      SVC_MARK(1)
      // 2. the tile's 64 samples (32 ring words) and the two in front of it -> v[64] in registers
      float v[kFT];
      {
        const uint32_t *tp = ring + (t & (kFRing2 - 1)) * kFStride2;
        const uint32_t *pp_ = (lane == 0) ? &lastdw[(g - 1) & 7] : ring + ((t - 1) & (kFRing2 - 1)) * kFStride2 + 31;
        uint4 row[8];
#pragma unroll
        for (int j = 0; j < 8; j++)
        {
          row[j] = reinterpret_cast<const uint4 *>(tp)[j];
        }
        uint32_t prevw = *pp_;                           // (generation 0, lane 0: whatever is there -- replaced below or history)
        // the rows go back to the stream waves: LDS has executed this wave's reads when it executes the flag's store
        const uint32_t lastw = (uint32_t)__builtin_amdgcn_readlane((int)row[7].w, ntl - 1);
        if (lane == 0)
        {
          lds_st(&lastdw[g & 7], lastw);
        }
        asm volatile("" ::: "memory");
        if (lane == 0)
        {
          lds_st(&ctl[kCtlRel], (uint32_t)g + 1u);
        }
        flow_hold_up(P, 9, g);
        constexpr bool kCoop = (HRFD_FLOW_COOP != 0) && SVC >= 2;
        if (kCoop && g + 1 == n_gens)
        {
          // the stream's last generation: its quarters are made by whoever is there (coop_quarters)
          if (lane == 0)
          {
            lds_st(&ctl[kCtlCoopOpen], (uint32_t)g + 1u);
          }
          coop_quarters<false>(g, theta_in, p_in);
          {
            const unsigned long long tw0 = __builtin_readcyclecounter();
            FlowSpin sp;
            while (lds_ld(&ctl[kCtlCoopDone]) != 4u && !sp.expired(P, ctl, fail_code, 12))
            {
              __builtin_amdgcn_s_sleep(1);
            }
            waited += __builtin_readcyclecounter() - tw0;
            lds_order();
          }
          const int r0 = t0 & (kFRing2 - 1);
          const uint32_t *vs = ring + ((r0 >= 384) ? 0 : r0 + 64) * kFStride2;
          const uint4 *sq = reinterpret_cast<const uint4 *>(vs + lane * kCoopStride);
#pragma unroll
          for (int i = 0; i < 16; i++)
          {
            const uint4 x4 = sq[i];
            v[4 * i] = u2f(x4.x);
            v[4 * i + 1] = u2f(x4.y);
            v[4 * i + 2] = u2f(x4.z);
            v[4 * i + 3] = u2f(x4.w);
          }
          th_last = u2f(vs[64 * kCoopStride + lane]);
          p_last = u2f(vs[64 * kCoopStride + 64 + lane]);
        }
        else if ((HRFD_FLOW_LASTTHETA != 0) && g + 1 == n_gens)
        {
          // the stream's last generation: the stream waves have left its thetas in the scratch area (they are in LDS with
          // the units' flags: a wave's stores are executed in order); what is left is wrap, gain and the filter's FIR half
          float thp, pp;
          theta_in_front(prevw, t == 0, theta_in, p_in, thp, pp);
          const int r0 = t0 & (kFRing2 - 1);
          const uint32_t *vs = ring + ((r0 >= 384) ? 0 : r0 + 64) * kFStride2;
          const uint4 *sq = reinterpret_cast<const uint4 *>(vs + lane * kCoopStride);
#pragma unroll
          for (int i = 0; i < 16; i++)
          {
            const uint4 x4 = sq[i];
            const float th[4] = {u2f(x4.x), u2f(x4.y), u2f(x4.z), u2f(x4.w)};
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
              const float pk = numerator_p<true>(th[k], thp, kgain);
              v[4 * i + k] = pk + pp;
              thp = th[k];
              pp = pk;
            }
          }
          th_last = thp;
          p_last = pp;
        }
        else
        {
          float thp, pp;
          theta_in_front(prevw, t == 0, theta_in, p_in, thp, pp);
#pragma unroll
          for (int j = 0; j < 8; j++)
          {
            const uint32_t w4[4] = {row[j].x, row[j].y, row[j].z, row[j].w};
            theta_to_v<1>(w4, thp, pp, *reinterpret_cast<float (*)[8]>(&v[8 * j]));
          }
          th_last = thp;
          p_last = pp;
        }
      }
      SVC_MARK(2)
      // 3. geometric partial sum of v over the own tile: P = sum_k c^k v[63 - k], c = -a1 (approximate on purpose)
      if (have && M > 0)
      {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const float cc = -a1, c2 = cc * cc;
        const f32x2 c22 = {c2, c2};
        f32x2 pab = {0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < kFT / 2; j++)
        {
          const f32x2 vv = {v[2 * j], v[2 * j + 1]};
          pab = __builtin_elementwise_fma(pab, c22, vv);
        }
        float p = __builtin_fmaf(pab.x, cc, pab.y);
        if (first && t == 0)
        {
          float yy = y_in;
          asm volatile("" : "+s"(yy));                   // (or the product is kept in a vector register for the whole stream: the bank kernel spilled it)
          p += deemph_pow(kFT) * yy;                     // the stream's past, as seen from the end of tile 0
        }
        parr[t & (kFPRing - 1)] = p;
      }
      lds_order();
      if (lane == 0)
      {
        lds_st(&pflag[g & 7], (uint32_t)g + 1u);
      }
      flow_hold_up(P, 4, g);
      SVC_MARK(3)
      if (g > 0 && M > 0)
      {
        const unsigned long long tw0 = __builtin_readcyclecounter();
        FlowSpin sp;
        while (lds_ld(&pflag[(g - 1) & 7]) != (uint32_t)g && !sp.expired(P, ctl, fail_code, 4))
        {
          __builtin_amdgcn_s_sleep(4);
        }
        waited += __builtin_readcyclecounter() - tw0;
        lds_order();
        if (fail_code != 0u)
        {
          break;
        }
      }
      SVC_MARK(4)
      // 4. seed, warm-up passes, tile
      const bool active = have && t >= fa_t;
      const bool exact0 = first && t == 0;               // the stream starts here: every pass starts from the carried y
      FlowTile o;
      o.y = 0.0f;
      o.sfirst0 = o.sfirst1 = o.slast0 = o.slast1 = 0u;
#pragma unroll
      for (int i = 0; i < 8; i++)
      {
        o.ud[i] = 0u;
      }
      float y_spec = 0.0f;
      {
        // y at the end of tile t - 1: sum_m (c^64)^(m-1) P[t - m], oldest first
        float y = 0.0f;
        if (M > 0)
        {
          float acc = 0.0f;
          for (int m = M; m >= 1; m--)
          {
            const int idx = t - m;
            const float pv = (idx >= 0) ? parr[idx & (kFPRing - 1)] : 0.0f;
            acc = __builtin_fmaf(acc, P.flow_seed_ct, pv);
          }
          y = acc;
        }
        y = exact0 ? y_in : y;
        for (int k = 0; k < wt; k++)
        {
          const float ye = flow_warm_reg(v, y);          // the lane's tile, from the start this pass was given
          // the last lane's value for lane 0 of the next generation, that generation's for ours
          if (ntl == 64 && lane == 63)
          {
            wcar[8 * k + (g & 7)] = ye;
          }
          asm volatile("" ::: "memory");
          if (ntl == 64 && lane == 63)
          {
            lds_st(&wflag[8 * k + (g & 7)], (uint32_t)g + 1u);
          }
          flow_hold_up(P, 10, g);
          float carry = 0.0f;
          if (g > 0)
          {
            const unsigned long long tw0 = __builtin_readcyclecounter();
            FlowSpin sp;
            while (lds_ld(&wflag[8 * k + ((g - 1) & 7)]) != (uint32_t)g && !sp.expired(P, ctl, fail_code, 11))
            {
              __builtin_amdgcn_s_sleep(2);
            }
            waited += __builtin_readcyclecounter() - tw0;
            lds_order();
            carry = wcar[8 * k + ((g - 1) & 7)];
          }
          y = u2f(shr1(f2u(ye), f2u(carry)));
          y = exact0 ? y_in : y;
        }
        y_spec = y;
        if (active && fail_code == 0u)
        {
          if (small_y)
          {
            flow_tile_reg<false>(v, y, o);
          }
          else
          {
            flow_tile_reg<true>(v, y, o);
          }
        }
      }
      if (fail_code != 0u)
      {
        break;
      }
      SVC_MARK(5)
      // 5. generations are verified in order ...
      {
        const unsigned long long tw0 = __builtin_readcyclecounter();
        FlowSpin sp;
        while (lds_ld(&ctl[1]) != (uint32_t)g && !sp.expired(P, ctl, fail_code, 5))
        {
          __builtin_amdgcn_s_sleep(2);
        }
        waited += __builtin_readcyclecounter() - tw0;
        lds_order();
      }
      if (fail_code != 0u)
      {
        break;
      }
      SVC_MARK(6)
      const uint32_t left_y = wfin[0], left_s0 = wfin[1], left_s1 = wfin[2];
      // 6. every lane but the first runnable one checks its speculated start against its left neighbour's end;
      //    a tile that has not merged is re-run from the true value, ascending (rare; v is still in the lane's registers)
      {
        const float y_left = u2f(shr1(f2u(o.y), left_y));
        const bool bad = active && t > fa_t && !same_trajectory(y_left, y_spec);
        unsigned long long bm = __ballot(bad);
        while (bm != 0ull)
        {
          const int l = __ffsll((long long)bm) - 1;      // wave-uniform
          bm &= ~(1ull << l);
          repairs++;
          const float y_true = u2f(shr1(f2u(o.y), left_y));
          if (lane == l)
          {
            if (small_y)
            {
              flow_tile_reg<false>(v, y_true, o);
            }
            else
            {
              flow_tile_reg<true>(v, y_true, o);
            }
          }
          // the right neighbour's speculation must now match the corrected final y
          const float y_new_left = u2f(shr1(f2u(o.y), left_y));
          const bool bad2 = (lane == l + 1) && active && !same_trajectory(y_new_left, y_spec);
          bm |= __ballot(bad2);
        }
      }
      // ... and hand their last lane's end to the next one at once
      const uint32_t fy = (uint32_t)__builtin_amdgcn_readlane((int)f2u(o.y), ntl - 1);
      const uint32_t fs0 = (uint32_t)__builtin_amdgcn_readlane((int)o.slast0, ntl - 1);
      const uint32_t fs1 = (uint32_t)__builtin_amdgcn_readlane((int)o.slast1, ntl - 1);
      if (lane == 0)
      {
        wfin[0] = fy;
        wfin[1] = fs0;
        wfin[2] = fs1;
      }
      asm volatile("" ::: "memory");
      if (lane == 0)
      {
        lds_st(&ctl[1], (uint32_t)g + 1u);
      }
      flow_hold_up(P, 5, g);
      // the integer stages follow in a chain of their own
      {
        const unsigned long long tw0 = __builtin_readcyclecounter();
        FlowSpin sp;
        while (lds_ld(&ctl[2]) != (uint32_t)g && !sp.expired(P, ctl, fail_code, 6))
        {
          __builtin_amdgcn_s_sleep(2);
        }
        waited += __builtin_readcyclecounter() - tw0;
        lds_order();
      }
      if (fail_code != 0u)
      {
        break;
      }
      // 7. U[0] of every tile: S[-4 .. -1] are the LEFT lane's last four samples (its final ones)
      {
        const uint32_t ls0 = shr1(o.slast0, left_s0), ls1 = shr1(o.slast1, left_s1);
        int acc = 1 << 14;
        acc = dot2(ls0, kRevWbD1.p[0], acc);
        acc = dot2(ls1, kRevWbD1.p[1], acc);
        acc = dot2(o.sfirst0, kRevWbD1.p[2], acc);
        acc = dot2(o.sfirst1, kRevWbD1.p[3], acc);
        o.ud[0] = (o.ud[0] & 0xffff0000u) | ((uint32_t)q15_out(acc) & 0xffffu);
      }
      if (have)
      {
        uint4 *up = reinterpret_cast<uint4 *>(uring + ((8 * t) & (kFUDw - 1)));
        up[0] = make_uint4(o.ud[0], o.ud[1], o.ud[2], o.ud[3]);
        up[1] = make_uint4(o.ud[4], o.ud[5], o.ud[6], o.ud[7]);
      }
      SVC_MARK(7)
      // 8. V[k] = D(12,4)(U), two per lane and pass (WbFmDemodulator.cc:478-486)
      for (int i = lane; i < 2 * ntl; i += 64)
      {
        const int k = 256 * g + 2 * i;                   // even
        const uint4 ua = *reinterpret_cast<const uint4 *>(uring + ((2 * k - 4) & (kFUDw - 1)));
        const uint4 ub = *reinterpret_cast<const uint4 *>(uring + ((2 * k) & (kFUDw - 1)));
        const uint32_t uu[8] = {ua.x, ua.y, ua.z, ua.w, ub.x, ub.y, ub.z, ub.w};   // U[4k-8 .. 4k+7]
        int acc0 = 1 << 14, acc1 = 1 << 14;
#pragma unroll
        for (int j = 0; j < 6; j++)
        {
          acc0 = dot2(uu[j], kRevD12.p[j], acc0);
          acc1 = dot2(uu[j + 2], kRevD12.p[j], acc1);
        }
        vring[(k >> 1) & (kFVDw - 1)] = ((uint32_t)q15_out(acc0) & 0xffffu) | ((uint32_t)q15_out(acc1) << 16);
      }
      // 9. PCM[p] = D(40,2)(V), two per lane (WbFmDemodulator.cc:488-496)
      if (have)
      {
        const int pp = 128 * g + 2 * lane;               // even; V[2pp-38 .. 2pp+3] = dwords pp-19 .. pp+1
        int acc0 = 1 << 14, acc1 = 1 << 14;
        uint32_t prev = vring[(pp - 19) & (kFVDw - 1)];
#pragma unroll
        for (int j = 0; j < 20; j++)
        {
          const uint32_t next = vring[(pp - 18 + j) & (kFVDw - 1)];
          acc0 = dot2(prev, kRevD40.p[j], acc0);
          acc1 = dot2(next, kRevD40.p[j], acc1);
          prev = next;
        }
        if (GATED)
        {
          // the stream is a list of blocks: PCM pair pp / 2 of the stream is pair (pp mod npcm) / 2 of block blist[pp / npcm]
          const uint32_t npcm = (uint32_t)n256 >> 5;
          const uint32_t k = (uint32_t)pp / npcm;
          pcm32[((uint32_t)blist[k & 63u] * npcm + ((uint32_t)pp - k * npcm)) >> 1] =
              ((uint32_t)q15_out(acc0) & 0xffffu) | ((uint32_t)q15_out(acc1) << 16);
        }
        else if ((uint32_t)pp >= pcm_off)
        {
          pcm32[((uint32_t)pp - pcm_off) >> 1] = ((uint32_t)q15_out(acc0) & 0xffffu) | ((uint32_t)q15_out(acc1) << 16);
        }
      }
      SVC_MARK(8)
      // 10. cross-block check values: y at block-relative position -705 = the end of the tile [-768, -704)
      //     (GATED: one exact stream from the committed state, nothing to check)
      if (have && !GATED)
      {
        const int x = 64 * t + 768 - hal;                // = (number of blocks completed) * n256 when this is such a tile
        if (x >= 0)
        {
          // (behind an opaque copy: left alone the compiler keeps the division's reciprocal -- and the lane's addresses of
          //  the end-of-stream block below -- in vector registers for the whole stream, which the tile's 64 do not leave room for)
          int nn = n256;
          asm volatile("" : "+s"(nn));
          const int q = x / nn;
          if (q * nn == x)
          {
            const uint32_t b = b_first + (uint32_t)q;    // the block this value stands in front of
            if (q == 0)
            {
              P.chk_spec[(size_t)c * P.n_blocks + b] = o.y;   // speculated by this run (b_first > 0)
            }
            else
            {
              P.chk_pub[(size_t)c * P.n_blocks + b - 1] = o.y;
              if (b < b_end)
              {
                P.chk_spec[(size_t)c * P.n_blocks + b] = o.y;
                if (local)
                {
                  lds_st(&finl[b & 63u], f2u(o.y));
                }
              }
            }
          }
        }
      }
      // 11. hand over to the next generation
      {
        if (g + 1 == n_gens && (GATED || b_end == P.n_blocks))
        {
          // the carried state for the next call (pending: committed when the launch verified clean)
          const uint32_t thl = (uint32_t)__builtin_amdgcn_readlane((int)f2u(th_last), ntl - 1);
          const uint32_t pl = (uint32_t)__builtin_amdgcn_readlane((int)f2u(p_last), ntl - 1);
          if (lane == 0)
          {
            so->wb_y = u2f(fy);
            so->wb_theta = u2f(thl);
            so->wb_p = u2f(pl);
            reinterpret_cast<uint32_t *>(so->wb_s)[0] = fs0;
            reinterpret_cast<uint32_t *>(so->wb_s)[1] = fs1;
          }
          int ln = lane;
          asm volatile("" : "+v"(ln));
          if (ln < 4)
          {
            reinterpret_cast<uint32_t *>(so->wb_u)[ln] = uring[(8 * n_tiles - 4 + ln) & (kFUDw - 1)];
          }
          if (ln < 19)
          {
            reinterpret_cast<uint32_t *>(so->wb_v)[ln] = vring[(2 * n_tiles - 19 + ln) & (kFVDw - 1)];
          }
          if (local)
          {
            // the same section in ChanState's order, dwords from wb_theta on: theta, p, y, pad, s (2), u (4), v (20)
            if (lane == 0)
            {
              lds_st(&finl[128], thl);
              lds_st(&finl[129], pl);
              lds_st(&finl[130], fy);
              lds_st(&finl[131], 0u);
              lds_st(&finl[132], fs0);
              lds_st(&finl[133], fs1);
            }
            if (ln < 4)
            {
              lds_st(&finl[134 + ln], uring[(8 * n_tiles - 4 + ln) & (kFUDw - 1)]);
            }
            if (ln < 20)
            {
              lds_st(&finl[138 + ln], ln < 19 ? vring[(2 * n_tiles - 19 + ln) & (kFVDw - 1)] : 0u);
            }
          }
        }
        asm volatile("" ::: "memory");
        if (lane == 0)
        {
          lds_st(&ctl[2], (uint32_t)g + 1u);
        }
      }
      flow_hold_up(P, 6, g);
      SVC_MARK(9)
    }
    // Out of generations: help with the stream's last one (coop_quarters), once its wave says that its rows can be read
    if ((HRFD_FLOW_COOP == 1) && SVC >= 2 && fail_code == 0u && n_gens > 0 && !(HRFD_ABLATE & 1024))   // (2: no helpers, bring-up)
    {
      FlowSpin sp;
      while (lds_ld(&ctl[kCtlCoopOpen]) != (uint32_t)n_gens && lds_ld(&ctl[kCtlCoopNext]) != 15u && !sp.expired(P, ctl, fail_code, 13))
      {
        __builtin_amdgcn_s_sleep(4);
      }
      lds_order();
      if (fail_code == 0u && lds_ld(&ctl[kCtlCoopOpen]) == (uint32_t)n_gens)
      {
        coop_quarters<true>(n_gens - 1, theta_in, p_in);
      }
    }
    FLOW_TIME_MAX(45)
#ifdef HRFD_FLOW_PROBE
    if (P.dbg != nullptr && lane == 0)
    {
      for (int i = 0; i < 10; i++)
      {
        atomicAdd(&P.dbg[(size_t)blockIdx.x * kDbgSlots + 32 + i], sprobe[i]);
      }
    }
#endif
    if (repairs != 0u && lane == 0)
    {
      atomicAdd(&P.counters[kCntRepair], repairs);
      atomicAdd(&P.sticky[kCntTotRepair], repairs);
    }
  }

  // ------------------------------------------------------------------------------------------------ finish
  __device__ __forceinline__ void finish()
  {
    // The last wave of the last workgroup of a channel finishes the channel (finish_channel: squelch tracker, checks
    // of both speculations, n_pcm / allowed outputs, commit of the pending state): no kernel behind this one.
    // Release / acquire at agent scope around the two counters (MI355X_MICROARCH, "Workgroup dispatch ... visibility").
    if (GATED)
    {
      // The exact pass over the allowed blocks is through.  The channel's last wave writes the squelch outputs of every
      // block, commits the pending state -- the demodulator's section when any block was demodulated (it is the state
      // behind the LAST ALLOWED block: a closed gate freezes it), the front end's 16 bytes and the tracker always --
      // and takes the channel's failure back: nothing is left for the host to replay.  A wait that expired keeps it.
      if (fail_code != 0u && lane == 0)
      {
        lds_st(&ctl[6], 1u);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the FIR modes' pending state went to memory: complete before the count)
      uint32_t prev = 0;
      if (lane == 0)
      {
        prev = atomicAdd(&ctl[5], 1u);
      }
      if (__builtin_amdgcn_readfirstlane((int)prev) == kWaves - 1)
      {
        lds_order();
        if (lds_ld(&ctl[6]) == 0u)
        {
          const uint32_t nb = P.n_blocks;
          const EpilogueParams &E = P.fin;
          ChanState *dst = P.state + c;
          // allowed[b]: b is in the list
          bool allowed = false;
          for (uint32_t k = 0; k < n_stream_blocks; k++)
          {
            allowed = allowed || (uint32_t)blist[k] == (uint32_t)lane;
          }
          if ((uint32_t)lane < nb)
          {
            const size_t ounit = (size_t)c * E.out_blocks + E.out_b0 + lane;
            if (E.allowed != nullptr)
            {
              E.allowed[ounit] = allowed ? 1 : 0;
            }
            if (E.n_pcm != nullptr)
            {
              E.n_pcm[ounit] = allowed ? E.n_pcm_per_block : 0u;
            }
          }
          if (lane < 4)
          {
            reinterpret_cast<uint32_t *>(dst->fe_tail)[lane] = lds_ld(&finl[162 + lane]);
          }
          if (kWb && n_stream_blocks != 0u && lane < 30)
          {
            reinterpret_cast<uint32_t *>(&dst->wb_theta)[lane] = lds_ld(&finl[128 + lane]);
          }
          if (!kWb && n_stream_blocks != 0u)
          {
            // the mode's section of the pending state, as the service waves left it in memory (written on this CU)
            int off, nd;
            state_section(cfg.mode, off, nd);
            const uint32_t *ssec = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(P.state_out + c) + off);
            uint32_t *dsec = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(dst) + off);
            for (int i = lane; i < nd; i += 64)
            {
              dsec[i] = ssec[i];
            }
          }
          if (lane == 0)
          {
            dst->tracking = lds_ld(&gctl[1]) != 0u ? 1u : 0u;
            E.chan_fail[c] = 0u;
            E.chan_poison[c] = 0u;
            atomicSub(&E.counters[kCntFail], 1u);
            atomicSub(&E.sticky[kCntTotViol], 1u);
            atomicAdd(&E.sticky[kCntTotGated], 1u);
          }
        }
      }
    }
    else if (local)
    {
      // the channel was this workgroup's alone: every input of the verdict is in LDS, nothing is read back from memory
      // (no wait for this wave's stores either: the end of the kernel is their fence)
      if (fail_code != 0u && lane == 0)
      {
        lds_st(&ctl[6], 1u);
      }
      uint32_t prev = 0;
      if (lane == 0)
      {
        prev = atomicAdd(&ctl[5], 1u);
      }
      if (__builtin_amdgcn_readfirstlane((int)prev) == kWaves - 1)
      {
        lds_order();
        const uint32_t nb = P.n_blocks;
        FinishIn<3> I;
        I.mode = 3;
        I.tracking = lds_ld(&finl[160]);
        I.poison = lds_ld(&finl[161]);
        I.expired = lds_ld(&ctl[6]);
        I.pres0 = ((uint32_t)lane < nb) ? (lds_ld(&blkout[lane]) >> 31) : 0u;
        I.pl_raw = lds_ld(&blkout[(nb - 1u) & 63u]) >> 31;
        I.pp_raw = lds_ld(&blkout[(nb - 2u) & 63u]) >> 31;  // n_blocks >= 2 here
        // one run: the value published in front of a block IS the one the block started from (a NaN fails, as ever)
        I.spec0 = (lane > 0 && (uint32_t)lane < nb) ? u2f(lds_ld(&finl[lane])) : 0.0f;
        I.pub0 = I.spec0;
        I.fe = (lane < 4) ? lds_ld(&finl[162 + lane]) : 0u;
        I.sec[0] = (lane < 30) ? lds_ld(&finl[128 + lane]) : 0u;
        finish_apply<3>(P.fin, c, lane, I);
      }
    }
    else if (P.self_finish)
    {
      if (fail_code != 0u && lane == 0)
      {
        P.fin.chan_expired[c] = 1u;
      }
      // this wave's global stores are done (they are in the XCD's L2, which every wave of this CU reads through) ...
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      uint32_t prev = 0;
      if (lane == 0)
      {
        prev = atomicAdd(&ctl[5], 1u);
      }
      if (__builtin_amdgcn_readfirstlane((int)prev) == kWaves - 1)
      {
        // ... and this is the last wave of the workgroup.  A channel cut into several runs is finished by the workgroup
        // that arrives last; what the others wrote comes from other CUs, possibly other XCDs: one agent-scope release
        // per workgroup in front of the arrival count, one acquire behind it (MI355X_MICROARCH, "Workgroup dispatch,
        // XCD placement & inter-workgroup visibility").  One run per channel (the usual case): nothing to fence.
        bool last = true;
        if (P.n_runs > 1u)
        {
          if ((uint32_t)(P.dbg_flags >> 16) == 8000u + run && ci == 0u)
          {
            for (int z = 0; z < 15; z++)
            {
              __builtin_amdgcn_s_sleep(127);
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          uint32_t arrived = 0;
          if (lane == 0)
          {
            arrived = __hip_atomic_fetch_add(&P.fin.chan_arrived[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          last = (uint32_t)__builtin_amdgcn_readfirstlane((int)arrived) == P.n_runs - 1u;
          if (last)
          {
            if (lane == 0)
            {
              __hip_atomic_store(&P.fin.chan_arrived[c], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero between launches
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
        }
        if (last)
        {
          finish_channel<kWb ? 3 : -1>(P.fin, c, lane);
        }
      }
    }
    FLOW_TIME_MAX(46)
    if (P.dbg != nullptr && lane == 0)
    {
      // per wave: cycles spent waiting (ring full / units not there yet / generation order); slot 0: the workgroup's cycles
      P.dbg[(size_t)blockIdx.x * kDbgSlots + 8 + wave] = waited;
      if (fail_code != 0u)
      {
        P.dbg[(size_t)blockIdx.x * kDbgSlots + 7] = 0x100000000ull * fail_code + (unsigned)wave + 1u;
      }
      if (tid == 0)
      {
        P.dbg[(size_t)blockIdx.x * kDbgSlots + 0] = __builtin_readcyclecounter() - t_kernel;
      }
    }
  }
};

template <int SVC_, bool GATED, bool DUMP, int MODE>
__device__ __forceinline__ void flow_body(const RxParams &P, uint32_t *const lds)
{
  // (the re-split WBFM chain gives its service waves a third of the work: more of them)
  constexpr int SVC = FlowLds<MODE>::kSplit ? HRFD_FLOW_SVC_WB : SVC_;
  Flow<SVC, GATED, DUMP, MODE> F(P, lds);
  if (!F.setup())
  {
    return;
  }
  if (F.wave >= SVC)
  {
    F.stream_waves();
  }
  else if constexpr (MODE != 3)
  {
    F.service_waves_fir();
  }
  else if constexpr (FlowLds<MODE>::kSplit)
  {
    F.service_waves_wbfm2();
  }
  else
  {
    F.service_waves_wbfm();
  }
  F.finish();
}

template <int SVC, bool GATED, bool DUMP, int MODE = 3>
__global__ __launch_bounds__(kThreads, 4) void k_rx_wbfm_flow(const RxParams P)
{
  __shared__ __attribute__((aligned(16))) uint32_t lds[FlowLds<MODE>::kTotal];
  flow_body<SVC, GATED, DUMP, MODE>(P, lds);
}

// A bank of several modes (BASELINE config 3) as ONE launch: one persistent workgroup per channel whatever its mode,
// the mode read from the channel's configuration -- no per-mode kernels, no streams to fork and join, no kernel
// boundaries inside a step.  Every workgroup holds a whole CU and runs for about the same time (the call's blocks of
// one channel), so a bank of as many channels as the chip has CUs is one round.
template <int SVC, bool DUMP = false>
__global__ __launch_bounds__(kThreads, 4) void k_rx_flow_bank(const RxParams P)
{
  constexpr int kDw = (FlowLds<3>::kTotal > FlowLds<2>::kTotal) ? ((FlowLds<3>::kTotal > FlowLds<14>::kTotal) ? FlowLds<3>::kTotal : FlowLds<14>::kTotal)
                                                                : ((FlowLds<2>::kTotal > FlowLds<14>::kTotal) ? FlowLds<2>::kTotal : FlowLds<14>::kTotal);
  __shared__ __attribute__((aligned(16))) uint32_t lds[kDw];
  uint32_t ci, run;
  if (!map_unit(blockIdx.x, P.n_list, P.n_runs, ci, run))
  {
    return;
  }
  const int mode = __builtin_amdgcn_readfirstlane(P.cfg[P.chan_list[ci]].mode);
  if (mode == 3)
  {
    flow_body<SVC, false, DUMP, 3>(P, lds);
  }
  else if (mode == 2)
  {
    flow_body<SVC, false, DUMP, 2>(P, lds);
  }
  else
  {
    flow_body<SVC, false, DUMP, 14>(P, lds);
  }
}

template __global__ void k_rx_flow_bank<HRFD_FLOW_SVC, false>(const RxParams);
template __global__ void k_rx_flow_bank<HRFD_FLOW_SVC, true>(const RxParams);
template __global__ void k_rx_wbfm_flow<HRFD_FLOW_SVC, false, true, 2>(const RxParams);
template __global__ void k_rx_wbfm_flow<HRFD_FLOW_SVC, false, true, 14>(const RxParams);
template __global__ void k_rx_wbfm_flow<HRFD_FLOW_SVC, false, false>(const RxParams);
template __global__ void k_rx_wbfm_flow<HRFD_FLOW_SVC, false, true>(const RxParams);
template __global__ void k_rx_wbfm_flow<HRFD_FLOW_SVC, true, false>(const RxParams);
template __global__ void k_rx_wbfm_flow<HRFD_FLOW_SVC, true, false, 2>(const RxParams);
template __global__ void k_rx_wbfm_flow<HRFD_FLOW_SVC, true, false, 14>(const RxParams);
template __global__ void k_rx_wbfm_flow<HRFD_FLOW_SVC, false, false, 2>(const RxParams);
template __global__ void k_rx_wbfm_flow<HRFD_FLOW_SVC, false, false, 14>(const RxParams);

} // namespace hrfd
