/* include/hrfd.h -- C ABI of libhrfd.so, the MI355X (gfx950) implementation of the
 * HackRfDiags demodulation / modulation hot path.
 *
 * Plain C: opaque handles, raw pointers, sizes, int return codes.  No C++ or
 * torch types cross this boundary.  Every entry point names the reference
 * interface it stands in for (file:line under /root/reference/radioDiags).
 * The reference-named C++ classes in hackrfdiags_amd/csrc/shim/ are thin
 * wrappers over these calls (INTEGRATION.md shows how they are linked in).
 *
 * Conventions
 *   - return 0 on success, a negative HRFD_E* code on failure
 *     (hrfd_last_error() returns a human-readable message for the calling thread)
 *   - "channel" = one independent IQ stream = one IqDataProcessor + its four
 *     demodulators in the reference (Radio.cc:164-203)
 *   - one "block" = 262144 bytes of interleaved int8 IQ at 2.048 MS/s = 64 ms
 *     (hackRf/hackrf.c:101, DataConsumer.h:15) -> 512 int16 PCM samples at 8 kS/s
 *   - handles are single-caller for process calls; setters may be called from
 *     another thread and take effect at the next process call (the reference
 *     has the same unsynchronised CLI-thread setters)
 */
#ifndef HRFD_H
#define HRFD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HRFD_VERSION 1

/* error codes */
#define HRFD_OK            0
#define HRFD_EINVAL       -1   /* bad argument (NULL, a size the reference cannot take either: odd, 0, larger than its arrays) */
#define HRFD_ENODEV       -2   /* no HIP device / HIP runtime failure */
#define HRFD_ENOMEM       -3
#define HRFD_ESTATE       -4   /* handle misuse */

/* demodulator modes == IqDataProcessor::demodulatorType (hdr_diags/IqDataProcessor.h:21) */
#define HRFD_MODE_NONE 0
#define HRFD_MODE_AM   1
#define HRFD_MODE_FM   2
#define HRFD_MODE_WBFM 3
#define HRFD_MODE_LSB  4
#define HRFD_MODE_USB  5

#define HRFD_ALL_CHANNELS 0xffffffffu

#define HRFD_BLOCK_BYTES   262144u   /* DATA_CONSUMER_BUFFER_SIZE, hdr_diags/DataConsumer.h:15 */
#define HRFD_PCM_PER_BLOCK 512u      /* PCM_BLOCK_SIZE, hdr_diags/BasebandDataProcessor.h:16 */

typedef struct hrfd_rx hrfd_rx;       /* C channels of IqDataProcessor + demodulators */
typedef struct hrfd_demod hrfd_demod; /* C channels of one {Am,Fm,WbFm,Ssb}Demodulator */
typedef struct hrfd_mod hrfd_mod;     /* C channels of SsbModulator / interpolateSignal */

const char *hrfd_last_error(void);
int hrfd_version(void);
/* number of visible HIP devices (0 when there is no GPU; never fails) */
int hrfd_device_count(void);

/* ------------------------------------------------------------------------------
 * Receive, outer boundary.
 * Replaces: IqDataProcessor::IqDataProcessor / ~IqDataProcessor
 *           (src_diags/IqDataProcessor.cc:56-160) for n_channels streams at once,
 *           together with the four demodulator objects Radio.cc:179-197 creates.
 * device < 0 selects the current HIP device.
 */
int hrfd_rx_create(uint32_t n_channels, int device, hrfd_rx **out);
int hrfd_rx_destroy(hrfd_rx *h);

/* IqDataProcessor::setDemodulatorMode (IqDataProcessor.cc:346-375); LSB/USB also
 * select the SSB demodulator's sideband, as the reference does. */
int hrfd_rx_set_mode(hrfd_rx *h, uint32_t channel, int mode);
/* {Am,Fm,WbFm,Ssb}Demodulator::setDemodulatorGain (e.g. WbFmDemodulator.cc:299) */
int hrfd_rx_set_gain(hrfd_rx *h, uint32_t channel, int mode, float gain);
/* IqDataProcessor::setSignalDetectThreshold (IqDataProcessor.cc:392-405), dBFS */
int hrfd_rx_set_threshold(hrfd_rx *h, uint32_t channel, int32_t threshold);
/* X::resetDemodulator (e.g. WbFmDemodulator.cc:265-278; NB: the WBFM de-emphasis
 * filter is not reset there, and is not reset here) */
int hrfd_rx_reset_demod(hrfd_rx *h, uint32_t channel, int mode);

/* IqDataProcessor::acceptIqData (IqDataProcessor.cc:926-1038) for every channel,
 * n_blocks consecutive blocks per channel in one call.  Host buffers:
 *   iq            [n_channels][n_blocks][block_bytes] int8, interleaved I,Q, 2.048 MS/s
 *   block_bytes   ANY even count, 2 .. 262144, like the reference: DataConsumer::acceptData clips longer buffers and
 *                 counts and PASSES ON shorter ones (DataConsumer.cc:229-241, :341-343; a USB transfer that ends early,
 *                 hackRf/hackrf.c:1443), and every decimator keeps its commutator position between calls
 *                 (Decimator_int16.cc:321-362).  What comes out per call is the reference's: the front end holds back
 *                 p = 0..7 IQ samples (hrfd_rx_pending_samples), a call completes floor((p + block_bytes/2) / 8) samples
 *                 at 256 kS/s, the Fs/4 rotation restarts at every call, the squelch mean is the call's own, the
 *                 demodulator emits what its three stages complete.  Odd counts are refused: the reference's Q loop then
 *                 reads bufferPtr[byteCount] (IqDataProcessor.cc:474) -- pass byteCount + 1 to get its result.  A call
 *                 too short to complete one 256 kS/s sample makes the reference divide by zero (SignalDetector.cc:255);
 *                 here it reports magnitude 0.
 *                 Speed: multiples of 1024 on a handle whose blocks all were multiples of 512 run on the streaming
 *                 kernels; other multiples of 512 (a transfer some USB packets short: 261632) pass through the exact
 *                 general kernel and the stream is back on the streaming kernels with the next full block; the first
 *                 block of any OTHER length moves the handle to the general kernel for good.
 *   gain_db       radio_adjustableReceiveGainInDb (Radio.cc:15) at call time
 *   pcm           [n_channels][n_blocks][hrfd_rx_pcm_capacity(block_bytes)] int16: a row holds n_pcm samples (out)
 *   n_pcm         [n_channels][n_blocks] samples actually produced (block_bytes/512 for whole multiples of 512 on an
 *                 open gate); 0 when squelched / mode NONE (the reference then makes no PCM callback)   (out)
 *   magnitude     [n_channels][n_blocks] Squelch::getSignalMagnitude() (out, may be NULL)
 *   signal_allowed[n_channels][n_blocks] Squelch::run() result       (out, may be NULL)
 *   iq256k_opt    [n_channels][n_blocks][hrfd_rx_iq256_capacity(block_bytes)] decimatedData after the Fs/4 mix -- what
 *                 `enable iqdump` sends by UDP; block_bytes/8 bytes for multiples of 16   (out, may be NULL)
 * Blocking: returns when the outputs are in the host buffers.
 */
int hrfd_rx_process_block(hrfd_rx *h, const int8_t *iq, uint32_t block_bytes,
                          uint32_t n_blocks, uint32_t gain_db, int16_t *pcm,
                          uint32_t *n_pcm, uint32_t *magnitude,
                          uint8_t *signal_allowed, int8_t *iq256k_opt);
/* Row lengths of the outputs above: ceil(block_bytes / 512) PCM samples, 2 * ceil(block_bytes / 16) bytes of the
 * 256 kS/s stream (what a call can complete at most, whatever the decimators hold). */
uint32_t hrfd_rx_pcm_capacity(uint32_t block_bytes);
uint32_t hrfd_rx_iq256_capacity(uint32_t block_bytes);
/* IQ samples the front end's three half-band decimators hold back after the calls so far (0..7, the same for every
 * channel of the handle; 0 while every block was a multiple of 16 bytes): the NEXT call of block_bytes bytes completes
 * floor((pending + block_bytes / 2) / 8) samples at 256 kS/s = IqDataProcessor::reduceSampleRate's return value / 2
 * (IqDataProcessor.cc:429-500).  Waits for the handle's last call. */
int hrfd_rx_pending_samples(hrfd_rx *h, uint32_t *pending);

/* IqDataProcessor::reduceSampleRate (IqDataProcessor.cc:429-500; public in the reference, not called by the
 * application): one block of every channel through the three half-band stages only -- the decimator pipelines advance,
 * the squelch and the demodulators are left alone.  iq [n_channels][block_bytes]; iq256k [n_channels][block_bytes/8]
 * receives the stream WITH the Fs/4 rotation (the front end is fused with the mixer here; the rotation is exactly
 * invertible, the shim class takes it out again).  Blocking; not to be called while another thread changes the
 * handle's modes (it parks them for the duration of the call). */
int hrfd_rx_reduce_sample_rate(hrfd_rx *h, const int8_t *iq, uint32_t block_bytes, int8_t *iq256k);

/* Same work with every buffer already resident in device memory (HBM); this is
 * the entry the batched benchmark drives.  d_iq is [n_channels][n_blocks]
 * [block_bytes] with channel_stride bytes between channels; block_bytes and the rows of the outputs as above
 * (d_n_pcm may be NULL; a d_iq256k_opt row is written up to the call's own count).  Asynchronous on
 * `stream` (a hipStream_t, NULL = the handle's own stream).  Optional outputs
 * may be NULL.  When n_blocks > 1 the call is one continuous stream per channel (or, for small banks and odd block
 * sizes, blocks demodulated concurrently): it speculates that every squelch gate in the batch is open and that the
 * WBFM de-emphasis tiles re-synchronise; both assumptions are verified on the device.  A gate that closes
 * inside the batch is repaired on the device as well: a gated pass behind the batch launch redoes the channels
 * concerned exactly (the stream of the blocks the squelch tracker allows).  hrfd_rx_sync() reports what is left.
 * Stream ordering is the caller's: the handle's own stream is non-blocking, so buffers that
 * were filled or cleared on another stream (e.g. a framework's default stream) must be
 * complete -- or `stream` must be that stream -- before this call.
 */
int hrfd_rx_process_device(hrfd_rx *h, const int8_t *d_iq, uint64_t channel_stride,
                           uint32_t block_bytes, uint32_t n_blocks, uint32_t gain_db,
                           int16_t *d_pcm, uint32_t *d_n_pcm, uint32_t *d_magnitude,
                           uint8_t *d_signal_allowed, int8_t *d_iq256k_opt,
                           void *stream);
/* Waits for the last hrfd_rx_process_device call.  *n_violations (may be NULL)
 * receives the number of CHANNELS that did not commit in that call (0 = every output is exact and every state
 * advanced).  The verdict is per channel: a channel that verified clean has advanced its state and
 * its outputs are exact; a failed channel has NOT advanced and the caller should
 * resubmit that channel one block per call (n_blocks == 1 is always exact) --
 * hrfd_rx_failed_channels says which, hrfd_rx_process_block does all of this by
 * itself.  Closed squelch gates are not failures (the device repairs them, see above) unless the batch has more
 * than 64 blocks or runs on the block kernels (small banks, odd block sizes).  (hrfd_rx_process_block runs a call of
 * more than 64 blocks as chunks of at most 64, each with the device's repair behind it: a caller of this entry that wants
 * the same keeps its calls at 64 blocks or fewer.) */
int hrfd_rx_sync(hrfd_rx *h, uint32_t *n_violations);
/* out[c] != 0 for the channels of that call that did not commit (n = n_channels). */
int hrfd_rx_failed_channels(hrfd_rx *h, uint8_t *out, uint32_t n);

/* ------------------------------------------------------------------------------
 * Receive, inner boundary: one demodulator class, n_channels instances.
 * Replaces: X::X(pcmCallbackPtr), X::acceptIqData(int8_t*,uint32_t),
 *           X::setDemodulatorGain, X::resetDemodulator,
 *           SsbDemodulator::set{Lsb,Usb}DemodulationMode
 *           (WbFmDemodulator.h:23-31, FmDemodulator.h:23-31, AmDemodulator.h:23-31,
 *            SsbDemodulator.h:24-34).
 * Input is 256 kS/s int8 IQ, already mixed; bytes_per_channel any even count <= 32768 (the reference's fixed member
 * arrays; its loops take any count and the decimators keep their positions, e.g. WbFmDemodulator.cc:395, :460-500).
 * pcm [n_channels][hrfd_demod_pcm_capacity(bytes_per_channel)], n_pcm [n_channels] (may be NULL) the samples produced:
 * bytes_per_channel / 64 for whole multiples of 64.  The PCM callback of the reference becomes the pcm/n_pcm output
 * pair; the C++ shim invokes the callback.
 */
int hrfd_demod_create(int mode, uint32_t n_channels, int device, hrfd_demod **out);
int hrfd_demod_destroy(hrfd_demod *h);
int hrfd_demod_reset(hrfd_demod *h, uint32_t channel);
int hrfd_demod_set_gain(hrfd_demod *h, uint32_t channel, float gain);
int hrfd_demod_set_sideband(hrfd_demod *h, uint32_t channel, int lsb);
int hrfd_demod_process(hrfd_demod *h, const int8_t *iq256k, uint32_t bytes_per_channel,
                       int16_t *pcm, uint32_t *n_pcm);
uint32_t hrfd_demod_pcm_capacity(uint32_t bytes_per_channel);   /* ceil(bytes_per_channel / 64) */

/* ------------------------------------------------------------------------------
 * Block transport in front of hrfd_rx.  Replaces the role of
 * DataConsumer (src_diags/DataConsumer.cc:219-262 acceptData: copy into a ring of messages and
 * queue; :319-351 the consumer thread calling IqDataProcessor::acceptIqData) for many channels:
 * a ring of n_slots pinned host batches [n_channels][n_blocks][block_bytes]; a submitted batch
 * is copied to the device, demodulated and its results copied back on separate streams, so the
 * transfer of one batch runs under the kernels of the previous one.  Results are the sequential
 * ones (a batch whose speculation fails is replayed exactly, with the batches in flight behind it).
 *   producer:  hrfd_ingest_acquire (pinned input buffer of the next free slot; HRFD_ESTATE when
 *              none is free) -> fill -> hrfd_ingest_submit (returns at once)
 *   consumer:  hrfd_ingest_collect (oldest submitted batch; blocks; pcm [C][B][hrfd_rx_pcm_capacity(block_bytes)],
 *              n_pcm / magnitude / signal_allowed [C][B]; pointers valid until that slot is
 *              acquired again)
 * The rx handle must not be used by other calls while batches are in flight.
 */
typedef struct hrfd_ingest hrfd_ingest;
int hrfd_ingest_create(hrfd_rx *rx, uint32_t block_bytes, uint32_t n_blocks, uint32_t n_slots,
                       hrfd_ingest **out);
int hrfd_ingest_destroy(hrfd_ingest *g);
int hrfd_ingest_acquire(hrfd_ingest *g, int8_t **iq_slot);
int hrfd_ingest_submit(hrfd_ingest *g, uint32_t gain_db);
int hrfd_ingest_collect(hrfd_ingest *g, const int16_t **pcm, const uint32_t **n_pcm,
                        const uint32_t **magnitude, const uint8_t **signal_allowed);
int hrfd_ingest_replayed(hrfd_ingest *g, uint64_t *n_batches);

/* ------------------------------------------------------------------------------
 * One host process, several devices.  The reference wires its whole receive path into one process
 * (src_diags/Radio.cc:164-237: one IqDataProcessor, its demodulators, one DataConsumer thread); a host that drives many
 * channels on the GPUs of a node stays one process too.  Channels share nothing, so n_devices devices are n_devices
 * contiguous channel shards -- shard g owns channels [first, first + count) of hrfd_fanout_channel_range: sizes differ
 * by at most one -- each an hrfd_rx of its own on its device, state pinned there.  No collective: IQ that lands on one
 * device leaves it as one peer copy per shard, every copy on the receiving shard's stream (xGMI is point to point: the
 * links out of the source work at the same time; in-process, no RCCL bootstrap), and the PCM comes back the same way.
 * devices[] may name a device more than once (several shards on one GPU).
 *   hrfd_fanout_scatter   d_iq_all [n_channels][n_blocks][block_bytes] on src_device -> the shards' input buffers;
 *                         src_stream: the stream that produced d_iq_all (awaited on the devices), NULL = complete
 *   hrfd_fanout_input     instead of scatter: the shard's own input buffer, for a host that feeds every device itself
 *   hrfd_fanout_process   IqDataProcessor::acceptIqData (IqDataProcessor.cc:926-1038) for every channel, n_blocks
 *                         blocks each, all devices at once, asynchronous
 *   hrfd_fanout_collect   waits, replays exactly what failed its speculation (as hrfd_rx_process_block does), gathers
 *                         pcm [n_channels][n_blocks][hrfd_rx_pcm_capacity(block_bytes)] and n_pcm [n_channels][n_blocks] (may be NULL)
 *                         into buffers on dst_device; *n_replayed (may be NULL) = channels replayed
 * The setters take channel numbers of the whole bank (HRFD_ALL_CHANNELS: every shard).
 * (The multi-process counterpart -- one rank per GPU, RCCL -- is hackrfdiags_amd/shard.py, used by bench.py --gpus N.)
 */
typedef struct hrfd_fanout hrfd_fanout;
int hrfd_fanout_channel_range(uint32_t n_channels, uint32_t n_shards, uint32_t shard, uint32_t *first, uint32_t *count);
int hrfd_fanout_create(uint32_t n_channels, const int *devices, uint32_t n_devices, hrfd_fanout **out);
int hrfd_fanout_destroy(hrfd_fanout *f);
int hrfd_fanout_shards(hrfd_fanout *f, uint32_t *n_shards);
int hrfd_fanout_set_mode(hrfd_fanout *f, uint32_t channel, int mode);
int hrfd_fanout_set_gain(hrfd_fanout *f, uint32_t channel, int mode, float gain);
int hrfd_fanout_set_threshold(hrfd_fanout *f, uint32_t channel, int32_t threshold);
int hrfd_fanout_scatter(hrfd_fanout *f, int src_device, const int8_t *d_iq_all, uint32_t block_bytes, uint32_t n_blocks,
                        void *src_stream);
int hrfd_fanout_input(hrfd_fanout *f, uint32_t shard, uint32_t block_bytes, uint32_t n_blocks, int8_t **d_iq,
                      uint32_t *first_channel, uint32_t *n_shard_channels);
int hrfd_fanout_process(hrfd_fanout *f, uint32_t gain_db);
int hrfd_fanout_collect(hrfd_fanout *f, int dst_device, int16_t *d_pcm_all, uint32_t *d_n_pcm_all, uint32_t *n_replayed);

/* ------------------------------------------------------------------------------
 * Transmit: PCM -> int8 IQ through the 8-stage x256 half-band interpolator.
 * kind HRFD_MOD_SSB replaces SsbModulator::acceptData (SsbModulator.cc:455-470)
 * incl. set{Lsb,Usb}ModulationMode / resetModulator (SsbModulator.h:23-35);
 * kind HRFD_MOD_INTERP replaces the signals/interpolateSignal tool
 * (signals/interpolateSignal.cc:250-374: int16 IQ pairs in, its own stage-1 table).
 */
#define HRFD_MOD_SSB    1
#define HRFD_MOD_INTERP 2
/* kinds HRFD_MOD_AM / HRFD_MOD_FM replace AmModulator::acceptData (AmModulator.cc:381-395,
 * modulateSignal :574-612) and FmModulator::acceptData (FmModulator.cc:393-407, modulateSignal
 * :586-627: an 8 kS/s Nco driven by deviation * pcm / 32768), same x256 cascade and tables;
 * PCM in, as for SSB.  FM goes through cosf / sinf: bit-exact on a glibc host since round 5 (hrfd_libm_variant() below). */
#define HRFD_MOD_AM     3
#define HRFD_MOD_FM     4
/* kind HRFD_MOD_WBFM replaces WbFmModulator::acceptData (WbFmModulator.cc:341-356: PCM x32,
 * a 256 kS/s Nco with runFast's table driven by deviation * x / 1024, x900, then x8): bit-exact
 * (the tables are built with the host's libm like the reference's). */
#define HRFD_MOD_WBFM   5
/* kinds HRFD_MOD_SIG_* replace the baseband generators of signals/ piped into interpolateSignal
 * (signals/makeThem.sh, generateBaseband.sh: `./a.out < pcm.raw | ./interpolateSignal > x.iq`):
 * am.cc:40-52 ((pcm*0.8 + 65536)/4 on both rails), dsb.cc:38-46 (pcm/4), pm.cc:41-53
 * (phase = pcm/60000*pi, 16000*cos/sin), fm.cc:44-77 (theta += pcm/65536*3.5, wrapped at +-2pi).
 * PCM in, int8 IQ at 2.048 MS/s out -- the .iq files that `load iqfile` plays (hrfd_play).
 * All four bit-exact (PM and FM go through cosf / sinf: on a glibc host, hrfd_libm_variant() below). */
#define HRFD_MOD_SIG_AM  6
#define HRFD_MOD_SIG_DSB 7
#define HRFD_MOD_SIG_PM  8
#define HRFD_MOD_SIG_FM  9
/* Which build of glibc's sinf / cosf the HOST's libm is: the reference reaches them through cos(float) / sin(float)
 * (Nco.cc:186-199, FmModulator.cc:600, signals/pm.cc, fm.cc) and the device restates that build bit for bit.
 * 1 = FMA build, 0 = without FMA, -1 = neither probed build (a libm that is not glibc's: the device follows the FMA
 * build and the FM modulator, Nco::run and the pm / fm generators may then differ from that host's libm by +-1 LSB). */
int hrfd_libm_variant(void);
int hrfd_mod_create(int kind, uint32_t n_channels, int device, hrfd_mod **out);
int hrfd_mod_destroy(hrfd_mod *h);
int hrfd_mod_reset(hrfd_mod *h, uint32_t channel);
int hrfd_mod_set_sideband(hrfd_mod *h, uint32_t channel, int lsb);
/* AmModulator::setModulationIndex (AmModulator.cc:329-336; default 0.8, accepted in [0, 1]) */
int hrfd_mod_set_modulation_index(hrfd_mod *h, uint32_t channel, float index);
/* FmModulator::setFrequencyDeviation (FmModulator.cc:336-346; default 3500 Hz) and
 * WbFmModulator::setFrequencyDeviation (WbFmModulator.cc:307-318; default 70000 Hz) */
int hrfd_mod_set_deviation(hrfd_mod *h, uint32_t channel, float deviation_hz);
/* pcm [n_channels][n_per_channel] int16 (SSB) or [n_channels][2*n_per_channel]
 * int16 IQ pairs (INTERP); iq_out [n_channels][512*n_per_channel] int8;
 * *out_bytes = 512*n_per_channel (bytes per channel, as the reference returns). */
int hrfd_mod_process(hrfd_mod *h, const int16_t *pcm, uint32_t n_per_channel,
                     int8_t *iq_out, uint32_t *out_bytes);
int hrfd_mod_process_device(hrfd_mod *h, const int16_t *d_pcm, uint32_t n_per_channel,
                            int8_t *d_iq_out, void *stream);
int hrfd_mod_sync(hrfd_mod *h);

/* ------------------------------------------------------------------------------
 * The transmit side's PCM ring, one per channel (host code).  Same slots, table
 * and pacing policy as BasebandDataProcessor (src_diags/BasebandDataProcessor.cc: ctor :41-84,
 * getNextUnfilledBuffer :410-425, getNextFilledBuffer :476-606 -- more than 10 blocks of lag
 * drops one, fewer than 6 sends the previous one again, not running reads zeros --, start/stop
 * :306-356).  hrfd_txring_read_batch gathers one 512-sample block per channel into
 * batch[n_channels][512], the input of hrfd_mod_process(h, batch, 512, ...).
 * stats: {produced, consumed, dropped, added, writer index, reader index}.
 */
typedef struct hrfd_txring hrfd_txring;
int hrfd_txring_create(uint32_t n_channels, hrfd_txring **out);
int hrfd_txring_destroy(hrfd_txring *r);
int hrfd_txring_set_running(hrfd_txring *r, uint32_t channel, int running);
int hrfd_txring_write(hrfd_txring *r, uint32_t channel, const int16_t *pcm512);
int hrfd_txring_read_batch(hrfd_txring *r, int16_t *batch);
int hrfd_txring_stats(hrfd_txring *r, uint32_t channel, uint32_t *out6);

/* ------------------------------------------------------------------------------
 * Cyclic playback of an .iq file (raw int8 IQ at 2.048 MS/s): DataProvider::loadIqFile /
 * getIqData (src_diags/DataProvider.cc:235-300, 122-131, 174-231) with the file image resident in
 * HBM and one read position per channel, so that one recording drives many receive channels
 * (SURVEY 8f rank 4).  hrfd_play_get_device fills d_out[c][bytes_per_channel] (channel_stride bytes
 * apart) from every channel's position and advances it modulo the file length, exactly like
 * retrieveIqDataFromBuffer; nothing is written while no file is loaded (as in the reference).
 */
typedef struct hrfd_play hrfd_play;
int hrfd_play_create(uint32_t n_channels, int device, hrfd_play **out);
int hrfd_play_destroy(hrfd_play *h);
int hrfd_play_load_file(hrfd_play *h, const char *path);
int hrfd_play_load(hrfd_play *h, const int8_t *bytes, uint32_t n_bytes);
int hrfd_play_set_position(hrfd_play *h, uint32_t channel, uint32_t byte_index);
int hrfd_play_get_position(hrfd_play *h, uint32_t channel, uint32_t *byte_index);
int hrfd_play_get_device(hrfd_play *h, int8_t *d_out, uint64_t channel_stride,
                         uint32_t bytes_per_channel, void *stream);
int hrfd_play_get(hrfd_play *h, int8_t *out, uint32_t bytes_per_channel);

/* ------------------------------------------------------------------------------
 * Nco (Nco/Nco.cc:186-257, Nco/PhaseAccumulator.cc:157-181): n_channels
 * oscillators advanced `count` samples each; fast != 0 selects runFast's table.
 * i_out/q_out are [n_channels][count] float host buffers.
 * Accuracy: the phase sequence is the reference's bit for bit (float accumulate, double-compare
 * wrap).  fast != 0 (Nco::runFast): the values are the host-built table's -- bit-exact.
 * fast == 0 (Nco::run): the reference calls libm sinf/cosf; the device evaluates cos/sin of the
 * same phase in double and rounds to float, so a value may differ from glibc's by one ulp (the
 * float-trig tolerance the north star allows; tests/test_gpu_tx_nco.py states it).  The same holds
 * for what is built on Nco::run: the FM modulator and the pm / fm generators (int8 IQ within
 * +-1 LSB, see hrfd_mod_create).
 */
typedef struct hrfd_nco hrfd_nco;
int hrfd_nco_create(uint32_t n_channels, float sample_rate, float frequency, int device,
                    hrfd_nco **out);
int hrfd_nco_destroy(hrfd_nco *h);
int hrfd_nco_set_frequency(hrfd_nco *h, uint32_t channel, float frequency);
int hrfd_nco_reset(hrfd_nco *h, uint32_t channel);
int hrfd_nco_run(hrfd_nco *h, int fast, uint32_t count, float *i_out, float *q_out);

/* ------------------------------------------------------------------------------
 * Introspection used by the tests: copy out the constant tables the kernels use.
 * name: "HB1","HB2","HB3","WBFM_D1","POST_D12","AUDIO_D40","FM_TUNER_D32","AM_D1",
 * "AM_D2","AM_D3","SSB_DELAY","SSB_HILBERT","INTERP_HB8","INTERP_HB3","INTERP_HB2",
 * "INTERP_HB1","INTERPSIG_S1" (Q15 taps).  Returns the tap count, 0 if unknown. */
int hrfd_q15_table(const char *name, int16_t *out, int cap);
/* host-built atan2 table [256][256] (float bits) and dBFS table [257] */
int hrfd_atan2_table(float *out);
int hrfd_dbfs_table(int32_t *out);

#ifdef __cplusplus
}
#endif

#endif /* HRFD_H */
