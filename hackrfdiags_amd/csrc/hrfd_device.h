// hackrfdiags_amd/csrc/hrfd_device.h -- structures shared by the host side of the
// C ABI (hrfd_api.hip) and the gfx950 kernels (hrfd_rx_kernels.hip, ...).
#ifndef HRFD_DEVICE_H
#define HRFD_DEVICE_H

#include <stdint.h>

namespace hrfd {

// ---- geometry of one launch unit ("channel-block") --------------------------
// One workgroup demodulates one block of one channel: n256 = block_bytes/16
// samples of the 256 kS/s stream.  For blocks after the first of a call the
// workgroup also re-derives `hal` samples of history from the tail of the
// previous block's raw input (time-parallelism along one channel).
#ifndef HRFD_THREADS
#define HRFD_THREADS 1024
#endif
#ifndef HRFD_MAXN256
#define HRFD_MAXN256 16384
#endif
constexpr int kThreads = HRFD_THREADS;        // 16 wave64 per workgroup (2 workgroups = 32 waves per CU)
constexpr int kWaves = kThreads / 64;
constexpr int kMaxN256 = HRFD_MAXN256;        // 262144-byte block
// de-emphasis recurrence (phase B of k_rx_wbfm): the block is cut into tiles of kTile samples,
// one per lane of the first four waves.  A lane starts kWarmTiles tiles early from an
// APPROXIMATE y (a truncated geometric sum over kSeedTerms tiles, "seed") and is verified
// bit for bit against its left neighbour.
constexpr int kTile = 70;                     // = 2 (mod 4): 64-bit LDS accesses, 32 lanes in 32 different 8-byte banks
constexpr int kBWaves = 4;                    // waves that run the recurrence (one per SIMD)
constexpr int kMaxTiles = 64 * kBWaves;
#ifndef HRFD_WARM_TILES
#define HRFD_WARM_TILES 3
#endif
constexpr int kWarmTiles = HRFD_WARM_TILES;   // warm-up = 210 samples behind a seed that is good to a few ulp
constexpr int kSeedTerms = 5;                 // 0.949^350 = 1.2e-8
constexpr int kWarm = 512;                    // argument of the test hook hrfd_rx_debug_set_warm that means "default"
constexpr int kHist = 704;                    // exact history kept in front of a block (>= 644)
constexpr int kNeedHist = 644;                // first history sample the integer stages read
// history re-derived in front of a speculative block: tile (kWarmTiles + kSeedTerms) must start at or
// before -(kNeedHist + 1), so -origin <= 645 + kTile - 1 + 8 * kTile = 1274
constexpr int kMaxHal = 1280;
constexpr int kKeepMax = (kWarmTiles + kSeedTerms + 1) * kTile;   // v history a continuation block restores (630)
constexpr int kMaxNV = kMaxN256 + kMaxHal;    // floats of the v/y stream in LDS
// arithmetic atan2 (theta_arith in hrfd_rx_kernels.hip)
constexpr int kTriEntries = 129 * 130 / 2;    // (a, b) with 0 <= b <= a <= 128
constexpr int kCorrBytes = 8448;              // kTriEntries padded to 16-byte copies
constexpr int kInvEntries = 132;              // 1/a for a = 0..128 (entry 0 is 0), padded
constexpr int kQuadRow = 129;                 // first-quadrant table (theta_quad): TQ[|q| * 129 + |i|], 0 <= |i|, |q| <= 128
constexpr int kQuadEntries = 129 * 129;
constexpr int kQuadDwords = 16644;            // padded to 16-byte copies

// carried history sizes of the integer stages ("carried state")
constexpr int kWbS = 4, kWbU = 8, kWbV = 38;  // WBFM: last N-M inputs of D(8,4), D(12,4), D(40,2)
constexpr int kFmTail = 704;                  // FM:  iq256 samples (>= 684)
constexpr int kAmTail = 384;                  // AM/SSB: iq256 samples (>= 324, multiple of 64)
constexpr int kSsbHist = 32;                  // 8 kS/s I/Q history (>= 30)

// ---- per-channel persistent state (device memory, one per channel) ----------
struct alignas(16) ChanState
{
  // front end: the last 16 raw input bytes (7-sample halo per rail, A2)
  int8_t fe_tail[16];
  // squelch: SignalTracker state (A5) -- 1 = Tracking
  uint32_t tracking;
  uint32_t pad0[3];

  // WBFM (W2/W3): theta of the last sample, b1*x[n-1], y[n-1], stage histories
  float wb_theta;
  float wb_p;
  float wb_y;
  float pad1;
  int16_t wb_s[kWbS];
  int16_t wb_u[kWbU];
  int16_t wb_v[kWbV + 2];

  // FM (F1-F4): last kFmTail samples of the 256 kS/s stream it consumed
  // (stored as offset-binary index bytes i,q), everything else is derived.
  uint8_t fm_tail[2 * kFmTail];
  // the two post-demodulation stages hold samples that were scaled with the gain
  // of their time, so their pipelines are carried, not re-derived (D(12,4): 8, D(40,2): 38)
  int16_t fm_u[kWbU];
  int16_t fm_v[kWbV + 2];

  // AM (M1/M2) and SSB (S1/S2): iq256 tails + 8 kS/s recurrences
  uint8_t am_tail[2 * kAmTail];
  float am_x1, am_y1;                          // dc-removal x[n-1], y[n-1]
  uint8_t ssb_tail[2 * kAmTail];
  float ssb_x1, ssb_y1;
  int16_t ssb_i[kSsbHist];                     // last 8 kS/s I samples (delay line)
  int16_t ssb_q[kSsbHist];                     // last 8 kS/s Q samples (Hilbert)
};

// ---- per-channel configuration (host mirror uploaded when dirty) ------------
struct alignas(16) ChanCfg
{
  int32_t mode;                                // HRFD_MODE_*
  int32_t threshold;                           // dBFS
  float gain_am, gain_fm, gain_wbfm, gain_ssb;
  int32_t lsb;                                 // SSB sideband
  int32_t pad;
};

// ---- launch parameters -------------------------------------------------------
struct EpilogueParams
{
  uint32_t n_channels;         // channels this launch finishes: all of them, or chan_list[0 .. n_channels)
  uint32_t n_blocks;
  uint32_t n_pcm_per_block;
  uint32_t out_blocks, out_b0; // layout of allowed / n_pcm (see RxParams)
  const ChanCfg *cfg;
  ChanState *state;
  const ChanState *state_out;
  const uint8_t *present;
  uint8_t *allowed;            // optional out
  uint32_t *n_pcm;             // optional out
  const float *chk_pub;
  const float *chk_spec;
  uint32_t *counters;          // kCnt* of THIS launch (kCntRepair .. kCntFail): one of two sets, used alternately
  uint32_t *sticky;            // the handle's counter block: kCntSticky.. (totals) live here
  uint32_t *next_local;        // the other set: the launch's first channel clears it for the next launch (no memset between launches)
  const uint32_t *chan_list;   // nullptr: channels 0 .. n_channels - 1
  uint32_t first_channel;      // the channel whose finisher does the launch's bookkeeping
  uint32_t *chan_fail;         // [channels] verdict of the latest launch per channel: 0 = committed, else kFail* bits
  uint32_t *chan_poison;       // [channels] sticky: the channel failed and the host has not repaired it yet -- it must not
                               //   commit in launches submitted behind the failed one either (pipelined submission)
  uint32_t *chan_expired;      // [channels] set by a kernel whose wait expired (k_rx_wbfm_flow), cleared by the finisher
  uint32_t *chan_arrived;      // [channels] workgroups of the channel that are through (k_rx_wbfm_flow finishes its own
                               //   channels: the last workgroup of a channel does); zero between launches
};

struct RxParams
{
  const int8_t *iq;            // [C][n_blocks][block_bytes]
  uint64_t ch_stride;          // bytes between channels
  uint32_t block_bytes;
  uint32_t n_blocks;
  uint32_t n256;               // block_bytes / 16
  int32_t ntiles;              // de-emphasis tiles of kTile samples; tile i = [origin + i*kTile, +kTile), the last ends at n256
  int32_t origin;              // start of tile 0 (<= 0, even)
  int32_t hal;                 // history samples re-derived in front of a run's first block (b > 0); >= -origin, multiple of 64
  int32_t warm_tiles;          // warm-up length in tiles (kWarmTiles; tests shrink it)
  int32_t seed_terms;          // tiles summed for a lane's approximate start (kSeedTerms; 0 = start from y = 0: tests)
  float seed_ct;               // (-a1)^kTile
  int32_t serial;              // 1: exact one-lane recurrence (replay path, n_blocks == 1)
  int32_t src256;              // 1: the input IS the 256 kS/s mixed stream (inner demodulator API):
                               //    2 bytes per sample, no front end, no squelch
  int32_t dbg_flags;           // timing experiments only (results are wrong when non-zero)
  int32_t stagger;             // start-up delay of odd dispatch layers, in units of s_sleep(127) (~8k cycles)
  uint32_t run_len, n_runs;    // k_rx_wbfm: consecutive blocks of a channel per workgroup; runs per channel
  uint32_t out_blocks;         // outputs are laid out [C][out_blocks][...]; this launch fills
  uint32_t out_b0;             //   blocks out_b0 .. out_b0 + n_blocks - 1 of that layout
  const uint32_t *chan_list;   // channels of this launch (all in the same mode)
  uint32_t n_list;
  uint32_t gain_db;
  ChanState *state;            // read at b == 0
  ChanState *state_out;        // written by the last block (== state when n_blocks == 1)
  const ChanCfg *cfg;
  int16_t *pcm;                // [C][n_blocks][n256/32]
  uint32_t *magnitude;         // [C][n_blocks] block-mean magnitude
  uint8_t *present;            // [C][n_blocks] detector result (before the tracker)
  int8_t *iq256;               // optional [C][n_blocks][2*n256]
  int16_t *ssb_iq;             // SSB scratch [C][n_blocks][2][n256/32]: 8 kS/s I and Q rails
  const float *atan2_lut;      // [256][256]
  const uint8_t *at_corr;      // arithmetic atan2: correction bytes [kCorrBytes] and 1/a [kInvEntries]
  const float *at_inv;
  const uint8_t *at_corr2;     // first-octant table atan2 (theta_tab, k_rx_wbfm_flow): correction bytes [kCorrBytes]
  const float *at_t0;          //   and T0 [kCorrBytes floats]
  const uint32_t *at_quad;     // first-quadrant table with embedded corrections (theta_quad) [kQuadDwords]
  const int32_t *dbfs;         // [257]
  float *chk_pub;              // [C][n_blocks] y at (n256 - kHist + 59) of this block
  float *chk_spec;             // [C][n_blocks] y at (-kHist + 59) as speculated by this block
  uint32_t *counters;          // kCnt* of this launch
  uint32_t *sticky;            // the handle's totals (kCntTotRepair ...)
  EpilogueParams fin;          // k_rx_wbfm_flow finishes its own channels (the launch's finish parameters) ...
  int32_t self_finish;         // ... when this is set (also k_rx_fir<FM> and k_rx_post: they finish their channels themselves)
  int32_t flow_hal;            // k_rx_wbfm_flow: history samples in front of a run that does not start the call (multiple of 512)
  float flow_seed_ct;          // k_rx_wbfm_flow: (-a1)^64
  unsigned long long *dbg;     // optional [grid][kDbgSlots] s_memtime stamps at phase boundaries (diagnostic builds of bench only)
};

// ---- any block length: the general-phase chain (k_rx_ragged, hrfd_rx_ragged.hip) ----------------
// The reference's decimators keep a commutator position between calls (Decimator_int16.cc:321-362:
// decimationBufferIndex), so IqDataProcessor::acceptIqData takes ANY byteCount (DataConsumer.cc:229-241
// passes short USB transfers on).  While every block of a handle has been a multiple of 512 bytes all those
// positions are 0 at every call boundary and ChanState (N - M inputs per stage) is the whole state.  The first
// block of another length takes the handle "off the grid": from then on a channel's state is this structure --
// every stage's last N - 1 inputs and its commutator position, exactly the reference's ring contents in effect.
constexpr int kRagHead = 40;                   // >= N - 1 of the longest stage (D(40,2))
struct RagQ15
{
  int16_t tail[kRagHead];                      // tail[0 .. N-2]: the last N - 1 inputs, oldest first
  int32_t phase;                               // inputs since the last output, 0 .. M-1
  int32_t pad;
};
struct RagWb { float theta, p, y, pad; RagQ15 d1, d2, d3; };                 // WbFmDemodulator: previousTheta, b1*x[n-1], y[n-1]
struct RagFm { float th[4]; RagQ15 ti, tq, d2, d3; };                          // FmDemodulator: theta[n-4 .. n-1] (the differentiator's pipeline)
struct RagAs { float x1, y1, pad0, pad1; RagQ15 s[2][3]; RagQ15 delay, hilbert; };   // Am / SsbDemodulator: dc-removal x[n-1], y[n-1]
struct alignas(16) RagState
{
  uint32_t valid;                              // built from ChanState by k_rag_expand
  uint32_t fe_phase;                           // front end: IQ samples since the last 256 kS/s output, 0 .. 7
  uint32_t pad[2];
  int8_t fe_raw[32];                           // the last 16 raw IQ samples (14 are read)
  RagWb wb;
  RagFm fm;
  RagAs am, ssb;
};

struct RagParams
{
  const int8_t *iq;            // [C][n_blocks][block_bytes]
  uint64_t ch_stride;
  uint32_t block_bytes;        // even
  uint32_t n_blocks;
  int32_t src256;              // the input IS the mixed 256 kS/s stream (inner demodulator API)
  int32_t offgrid;             // 0: state in ChanState (every commutator at 0), 1: state in RagState
  uint32_t pcm_cap;            // samples per block row of pcm: ceil(block_bytes / 512) (src256: / 64)
  uint32_t iq256_cap;          // bytes per block row of iq256: 2 * ceil(block_bytes / 16)
  uint32_t out_blocks, out_b0; // layout of the outputs, as RxParams
  const uint32_t *chan_list;   // nullptr: channels 0 .. n_list - 1
  uint32_t n_list;
  uint32_t gain_db;
  ChanState *state;
  RagState *rag;
  const ChanCfg *cfg;
  int16_t *pcm;
  uint32_t *n_pcm;             // optional
  uint32_t *magnitude;
  uint8_t *allowed;            // optional
  int8_t *iq256;               // optional
  const float *atan2_lut;
  const int32_t *dbfs;
  // a launch's bookkeeping, the protocol of finish_apply (this kernel finishes its own channels and never speculates)
  uint32_t *counters, *sticky, *next_local;
  uint32_t first_channel;
  uint32_t *chan_fail, *chan_poison;
};

constexpr uint32_t kFailGate = 1u, kFailSpec = 2u, kFailPoison = 4u, kFailExpired = 8u;

constexpr int kCntRepair = 0;  // de-emphasis tiles re-run in place because their warm-up had not re-synchronised
constexpr int kCntGate = 1;    // blocks b > 0 whose squelch gate turned out closed
constexpr int kCntSpec = 2;    // blocks b > 0 whose first tile disagrees with the predecessor
constexpr int kCntFail = 3;    // channels of this launch whose state was NOT committed (hrfd_rx_debug_counters shows 1 = all committed here)
constexpr int kCntCommit = 3;
constexpr int kCntSticky = 4;  // counters from here on are never reset (totals since creation):
constexpr int kCntTotRepair = 4;
constexpr int kCntTotViol = 5; // channel-launches whose state was NOT committed
constexpr int kCntTotLaunch = 6;
constexpr int kCntScratch = 7;  // k_build_atan_corr's violation count (hrfd_rx_create only)
constexpr int kNumCounters = 8;   // counters visible through hrfd_rx_debug_counters
constexpr int kCntTotGated = 8;  // channel-launches redone on the device by the gated pass (k_rx_wbfm_flow<GATED>)
constexpr int kNumDevCounters = 10;
constexpr int kDbgSlots = 48;      // RxParams::dbg: per workgroup 0..5 phase stamps of thread 0, 6 placement, 8..23 per wave, 24..31 recurrence / stream-loop probes, 32..47 service-loop probes (k_rx_wbfm_flow)

} // namespace hrfd

#endif // HRFD_DEVICE_H
