// hackrfdiags_amd/csrc/hrfd_ingest.hip -- host-side block transport in front of hrfd_rx
// (SURVEY 8f rank 2).
//
// The reference moves every 262144-byte block through DataConsumer: acceptData()
// copies it into one of a ring of messages and queues it (DataConsumer.cc:219-262), the
// consumer thread dequeues and calls IqDataProcessor::acceptIqData (:319-351).  Here the
// ring holds BATCHES -- [n_channels][n_blocks][block_bytes] of int8 IQ in pinned host
// memory -- and the consumer is the GPU: a submitted slot is copied to the device on a copy
// stream, demodulated on the rx handle's stream (hrfd_rx_process_device) and its PCM /
// n_pcm / magnitude / gate results are copied back to pinned memory on a third stream, so
// that the transfer of batch k+1 runs under the kernels of batch k.
//
//   producer:  hrfd_ingest_acquire -> fill the slot -> hrfd_ingest_submit
//   consumer:  hrfd_ingest_collect  (oldest submitted batch, blocks until it is there)
//
// A multi-block batch is speculative (hrfd.h: hrfd_rx_process_device), per channel.  A channel that
// fails its checks does not commit its state, and neither does it in any batch launched behind that
// one (the device keeps a sticky flag per channel, chan_poison); collect() then replays THOSE
// CHANNELS of that batch and of the ones already in flight behind it through the exact path, in
// order, from the batches' device-resident inputs.  Results are always the sequential ones.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

struct hrfd_ingest
{
  hrfd_rx *rx = nullptr;
  uint32_t block_bytes = 0, n_blocks = 0, n_slots = 0, C = 0;
  size_t iq_bytes = 0, units = 0, pcm_elems = 0;
  hipStream_t s_in = nullptr, s_out = nullptr;

  struct Slot
  {
    int8_t *h_iq = nullptr, *d_iq = nullptr;
    int16_t *h_pcm = nullptr, *d_pcm = nullptr;
    uint32_t *h_npcm = nullptr, *d_npcm = nullptr, *h_mag = nullptr, *d_mag = nullptr;
    uint8_t *h_allowed = nullptr, *d_allowed = nullptr;
    uint32_t *h_counters = nullptr;                  // the launch's counters (kCntFail: channels that did not commit)
    uint32_t *h_fail = nullptr;                      // ... and which (chan_fail)
    hipEvent_t e_in = nullptr, e_comp = nullptr, e_out = nullptr;
    uint32_t gain_db = 0;
    int state = 0;                                   // 0 free, 1 acquired, 2 submitted, 3 collected (results in use)
  };
  std::vector<Slot> slots;
  uint32_t head = 0;                                 // next slot to acquire
  uint32_t tail = 0;                                 // oldest submitted slot
  uint32_t in_flight = 0;
  uint64_t replayed_batches = 0;
};

namespace {

int ingest_free(hrfd_ingest *g)
{
  if (g == nullptr)
  {
    return HRFD_OK;
  }
  (void)hipSetDevice(g->rx->device);
  (void)hipDeviceSynchronize();
  for (auto &sl : g->slots)
  {
    void *hostp[] = {sl.h_iq, sl.h_pcm, sl.h_npcm, sl.h_mag, sl.h_allowed, sl.h_counters, sl.h_fail};
    for (void *p : hostp)
    {
      if (p) (void)hipHostFree(p);
    }
    void *devp[] = {sl.d_iq, sl.d_pcm, sl.d_npcm, sl.d_mag, sl.d_allowed};
    for (void *p : devp)
    {
      if (p) (void)hipFree(p);
    }
    hipEvent_t ev[] = {sl.e_in, sl.e_comp, sl.e_out};
    for (hipEvent_t e : ev)
    {
      if (e) (void)hipEventDestroy(e);
    }
  }
  if (g->s_in) (void)hipStreamDestroy(g->s_in);
  if (g->s_out) (void)hipStreamDestroy(g->s_out);
  delete g;
  return HRFD_OK;
}

}  // namespace

extern "C" int hrfd_ingest_create(hrfd_rx *rx, uint32_t block_bytes, uint32_t n_blocks, uint32_t n_slots,
                                  hrfd_ingest **out)
{
  if (rx == nullptr || out == nullptr || n_blocks == 0 || n_slots < 2 || n_slots > 16 || block_bytes == 0 ||
      (block_bytes & 1u) != 0 || block_bytes > HRFD_BLOCK_BYTES)
  {
    return fail(HRFD_EINVAL, "hrfd_ingest_create: need a handle, an even block_bytes <= %u, "
                             "n_blocks > 0, 2..16 slots", HRFD_BLOCK_BYTES);
  }
  *out = nullptr;
  HIP_TRY(hipSetDevice(rx->device));
  hrfd_ingest *g = new hrfd_ingest;
  g->rx = rx;
  g->block_bytes = block_bytes;
  g->n_blocks = n_blocks;
  g->n_slots = n_slots;
  g->C = rx->n_channels;
  g->units = (size_t)g->C * n_blocks;
  g->iq_bytes = g->units * block_bytes;
  g->pcm_elems = g->units * ((block_bytes + 511u) / 512u);   // rows of hrfd_rx_pcm_capacity(block_bytes) samples
  g->slots.resize(n_slots);
  hipError_t e = hipStreamCreateWithFlags(&g->s_in, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->s_out, hipStreamNonBlocking);
  for (auto &sl : g->slots)
  {
    if (e == hipSuccess) e = hipHostMalloc((void **)&sl.h_iq, g->iq_bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&sl.h_pcm, g->pcm_elems * 2, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&sl.h_npcm, g->units * 4, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&sl.h_mag, g->units * 4, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&sl.h_allowed, g->units, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&sl.h_counters, sizeof(uint32_t) * kNumCounters, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&sl.h_fail, sizeof(uint32_t) * g->C, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void **)&sl.d_iq, g->iq_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&sl.d_pcm, g->pcm_elems * 2);
    if (e == hipSuccess) e = hipMalloc((void **)&sl.d_npcm, g->units * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&sl.d_mag, g->units * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&sl.d_allowed, g->units);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.e_in, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.e_comp, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.e_out, hipEventDisableTiming);
  }
  if (e != hipSuccess)
  {
    const int rc = fail(HRFD_ENOMEM, "hrfd_ingest_create: %s", hipGetErrorString(e));
    ingest_free(g);
    return rc;
  }
  *out = g;
  return HRFD_OK;
}

extern "C" int hrfd_ingest_destroy(hrfd_ingest *g) { return ingest_free(g); }

// The next free slot's pinned input buffer, [n_channels][n_blocks][block_bytes].  HRFD_ESTATE
// when every slot is submitted or still held by the consumer (collect first).
extern "C" int hrfd_ingest_acquire(hrfd_ingest *g, int8_t **iq_slot)
{
  if (g == nullptr || iq_slot == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_ingest_acquire: NULL");
  }
  hrfd_ingest::Slot &sl = g->slots[g->head];
  if (sl.state == 3)
  {
    sl.state = 0;                                    // results of an old collect are released by re-acquiring
  }
  if (sl.state != 0)
  {
    return fail(HRFD_ESTATE, "hrfd_ingest_acquire: no free slot (collect a batch first)");
  }
  sl.state = 1;
  *iq_slot = sl.h_iq;
  return HRFD_OK;
}

// Enqueue the acquired slot: H2D on the copy stream, demodulation on the rx stream, D2H of the
// results on the return stream.  Returns immediately.
extern "C" int hrfd_ingest_submit(hrfd_ingest *g, uint32_t gain_db)
{
  if (g == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_ingest_submit: NULL");
  }
  hrfd_ingest::Slot &sl = g->slots[g->head];
  if (sl.state != 1)
  {
    return fail(HRFD_ESTATE, "hrfd_ingest_submit: acquire a slot first");
  }
  hrfd_rx *rx = g->rx;
  HIP_TRY(hipSetDevice(rx->device));
  hipStream_t cs = rx->stream;
  sl.gain_db = gain_db;
  HIP_TRY(hipMemcpyAsync(sl.d_iq, sl.h_iq, g->iq_bytes, hipMemcpyHostToDevice, g->s_in));
  HIP_TRY(hipEventRecord(sl.e_in, g->s_in));
  HIP_TRY(hipStreamWaitEvent(cs, sl.e_in, 0));
  HIP_TRY(hipMemsetAsync(sl.d_pcm, 0, g->pcm_elems * 2, cs));      // squelched units: zeros, not stale PCM
  int rc = hrfd_rx_process_device(rx, sl.d_iq, (uint64_t)g->block_bytes * g->n_blocks, g->block_bytes, g->n_blocks,
                                  gain_db, sl.d_pcm, sl.d_npcm, sl.d_mag, sl.d_allowed, nullptr, cs);
  if (rc != HRFD_OK)
  {
    return rc;
  }
  HIP_TRY(hipMemcpyAsync(sl.h_counters, rx->d_local, sizeof(uint32_t) * kCntSticky, hipMemcpyDeviceToHost, cs));   // this launch's set
  HIP_TRY(hipMemcpyAsync(sl.h_fail, rx->d_chan, sizeof(uint32_t) * g->C, hipMemcpyDeviceToHost, cs));
  HIP_TRY(hipEventRecord(sl.e_comp, cs));
  HIP_TRY(hipStreamWaitEvent(g->s_out, sl.e_comp, 0));
  HIP_TRY(hipMemcpyAsync(sl.h_pcm, sl.d_pcm, g->pcm_elems * 2, hipMemcpyDeviceToHost, g->s_out));
  HIP_TRY(hipMemcpyAsync(sl.h_npcm, sl.d_npcm, g->units * 4, hipMemcpyDeviceToHost, g->s_out));
  HIP_TRY(hipMemcpyAsync(sl.h_mag, sl.d_mag, g->units * 4, hipMemcpyDeviceToHost, g->s_out));
  HIP_TRY(hipMemcpyAsync(sl.h_allowed, sl.d_allowed, g->units, hipMemcpyDeviceToHost, g->s_out));
  HIP_TRY(hipEventRecord(sl.e_out, g->s_out));
  sl.state = 2;
  g->head = (g->head + 1) % g->n_slots;
  g->in_flight++;
  return HRFD_OK;
}

// The oldest submitted batch: blocks until its results are in pinned memory.  The pointers stay
// valid until that slot is acquired again.  Any of the out pointers may be NULL.
extern "C" int hrfd_ingest_collect(hrfd_ingest *g, const int16_t **pcm, const uint32_t **n_pcm,
                                   const uint32_t **magnitude, const uint8_t **signal_allowed)
{
  if (g == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_ingest_collect: NULL");
  }
  if (g->in_flight == 0)
  {
    return fail(HRFD_ESTATE, "hrfd_ingest_collect: nothing submitted");
  }
  hrfd_rx *rx = g->rx;
  HIP_TRY(hipSetDevice(rx->device));
  hrfd_ingest::Slot &sl = g->slots[g->tail];
  HIP_TRY(hipEventSynchronize(sl.e_out));
  if (sl.h_counters[kCntFail] != 0)
  {
    // Some channels of this batch did not commit (their own checks failed, or they ran behind a failure of
    // theirs): they have not committed in anything launched since either.  Drain, then replay those channels of
    // this batch and of everything behind it, in order, through the exact path.
    HIP_TRY(hipStreamSynchronize(rx->stream));
    HIP_TRY(hipStreamSynchronize(g->s_out));
    HIP_TRY(hipMemset(rx->d_chan + rx->n_channels, 0, sizeof(uint32_t) * rx->n_channels));   // chan_poison
    uint32_t k = g->tail;
    for (uint32_t i = 0; i < g->in_flight; i++, k = (k + 1) % g->n_slots)
    {
      hrfd_ingest::Slot &r = g->slots[k];
      std::vector<uint32_t> redo;
      for (uint32_t c = 0; c < g->C; c++)
      {
        if (r.h_fail[c] != 0) redo.push_back(c);
      }
      if (redo.empty())
      {
        continue;
      }
      const int rc = rx_replay(rx, redo, r.d_iq, (uint64_t)g->block_bytes * g->n_blocks, g->block_bytes, g->n_blocks,
                               r.gain_db, r.d_pcm, r.d_npcm, r.d_mag, r.d_allowed, nullptr, rx->stream, false);
      if (rc != HRFD_OK)
      {
        return rc;
      }
      hipStream_t cs = rx->stream;
      HIP_TRY(hipMemcpyAsync(r.h_pcm, r.d_pcm, g->pcm_elems * 2, hipMemcpyDeviceToHost, cs));
      HIP_TRY(hipMemcpyAsync(r.h_npcm, r.d_npcm, g->units * 4, hipMemcpyDeviceToHost, cs));
      HIP_TRY(hipMemcpyAsync(r.h_mag, r.d_mag, g->units * 4, hipMemcpyDeviceToHost, cs));
      HIP_TRY(hipMemcpyAsync(r.h_allowed, r.d_allowed, g->units, hipMemcpyDeviceToHost, cs));
      HIP_TRY(hipStreamSynchronize(cs));
      r.h_counters[kCntFail] = 0;
      memset(r.h_fail, 0, sizeof(uint32_t) * g->C);
      HIP_TRY(hipEventRecord(r.e_out, g->s_out));    // already complete
      g->replayed_batches++;
    }
  }
  if (pcm != nullptr) *pcm = sl.h_pcm;
  if (n_pcm != nullptr) *n_pcm = sl.h_npcm;
  if (magnitude != nullptr) *magnitude = sl.h_mag;
  if (signal_allowed != nullptr) *signal_allowed = sl.h_allowed;
  sl.state = 3;
  g->tail = (g->tail + 1) % g->n_slots;
  g->in_flight--;
  return HRFD_OK;
}

// diagnostics: batches that had to be replayed through the exact path since creation
extern "C" int hrfd_ingest_replayed(hrfd_ingest *g, uint64_t *n)
{
  if (g == nullptr || n == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_ingest_replayed: NULL");
  }
  *n = g->replayed_batches;
  return HRFD_OK;
}
