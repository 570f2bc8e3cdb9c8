// hackrfdiags_amd/csrc/hrfd_fanout.hip -- one host process, N devices: the multi-GPU layer for a C++ host
// (host code only, no kernels of its own).
//
// The reference wires every object of the receive path into ONE process (Radio.cc:164-237: one IqDataProcessor, its
// four demodulators, one DataConsumer thread), and a host that links libhrfd in its place stays one process when it
// drives many channels on several GPUs.  Channels share nothing, so N devices are N contiguous channel shards --
// shard g of G owns channels [g C/G, (g+1) C/G), the first C mod G shards one more -- each an hrfd_rx of its own with
// its per-channel state pinned to its device for the life of the stream.  There is no collective: when all IQ lands
// on one device (the north star's "per-channel scatter") the shards leave it as one hipMemcpyPeerAsync per peer, each
// on the RECEIVING shard's stream -- xGMI is point to point, the 7 links out of the source carry their shards at the
// same time, and an in-process fan-out needs no RCCL bootstrap -- straight into the shard's persistent input buffer;
// the PCM (1 KiB per channel-block) comes back the same way.  (The multi-process counterpart, one rank per GPU over
// RCCL, is hackrfdiags_amd/shard.py: same shards, grouped ncclSend/ncclRecv instead of peer copies.)
//
//   hrfd_fanout_scatter   source buffer [C][B][block_bytes] on one device -> every shard's input buffer
//   hrfd_fanout_process   every shard demodulates its buffer (asynchronous, all devices at once)
//   hrfd_fanout_collect   per shard: wait, replay what failed its speculation exactly, PCM / n_pcm to the destination
//
// Several shards may name the same device (a one-GPU box rehearses the whole path that way: the tests do).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

struct hrfd_fanout
{
  struct Shard
  {
    int device = 0;
    uint32_t first = 0, count = 0;                       // channels [first, first + count)
    hrfd_rx *rx = nullptr;
    int8_t *d_iq = nullptr;                              // [count][n_blocks][block_bytes], grown on demand
    int16_t *d_pcm = nullptr;
    uint32_t *d_npcm = nullptr;
    size_t cap_iq = 0, cap_pcm = 0, cap_npcm = 0;
  };
  // "the source data is complete": one event per source device seen so far, created ON that device (an event is recorded
  // on a stream of its own device; any device's stream may wait for it)
  struct SrcEvent
  {
    int device;
    hipEvent_t ev;
  };
  std::vector<SrcEvent> src_events;
  uint32_t n_channels = 0;
  std::vector<Shard> shards;
  uint32_t block_bytes = 0, n_blocks = 0, gain_db = 0;   // of the batch in flight
  bool in_flight = false;
};

// Contiguous shards, sizes differing by at most one: shard g of G gets channels [lo, lo + n).
extern "C" int hrfd_fanout_channel_range(uint32_t n_channels, uint32_t n_shards, uint32_t shard, uint32_t *first,
                                         uint32_t *count)
{
  if (n_shards == 0 || shard >= n_shards || first == nullptr || count == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_fanout_channel_range: need shard < n_shards and result pointers");
  }
  const uint32_t base = n_channels / n_shards, extra = n_channels % n_shards;
  *first = shard * base + std::min(shard, extra);
  *count = base + (shard < extra ? 1u : 0u);
  return HRFD_OK;
}

static int fanout_free(hrfd_fanout *f)
{
  if (f == nullptr)
  {
    return HRFD_OK;
  }
  for (hrfd_fanout::Shard &s : f->shards)
  {
    (void)hipSetDevice(s.device);
    if (s.rx != nullptr && s.rx->stream != nullptr)
    {
      (void)hipStreamSynchronize(s.rx->stream);
    }
    if (s.d_iq) (void)hipFree(s.d_iq);
    if (s.d_pcm) (void)hipFree(s.d_pcm);
    if (s.d_npcm) (void)hipFree(s.d_npcm);
    rx_free(s.rx);
  }
  for (hrfd_fanout::SrcEvent &e : f->src_events)
  {
    (void)hipSetDevice(e.device);
    (void)hipEventDestroy(e.ev);
  }
  delete f;
  return HRFD_OK;
}

extern "C" int hrfd_fanout_create(uint32_t n_channels, const int *devices, uint32_t n_devices, hrfd_fanout **out)
{
  if (out == nullptr || devices == nullptr || n_devices == 0 || n_channels < n_devices)
  {
    return fail(HRFD_EINVAL, "hrfd_fanout_create: need a device list and at least one channel per shard");
  }
  *out = nullptr;
  const int visible = hrfd_device_count();
  if (visible <= 0)
  {
    return fail(HRFD_ENODEV, "hrfd_fanout_create: no HIP device visible (this library has no CPU path)");
  }
  hrfd_fanout *f = new hrfd_fanout;
  f->n_channels = n_channels;
  f->shards.resize(n_devices);
  for (uint32_t g = 0; g < n_devices; g++)
  {
    hrfd_fanout::Shard &s = f->shards[g];
    if (devices[g] < 0 || devices[g] >= visible)
    {
      const int rc = fail(HRFD_EINVAL, "hrfd_fanout_create: device %d is not one of the %d visible", devices[g], visible);
      fanout_free(f);
      return rc;
    }
    s.device = devices[g];
    (void)hrfd_fanout_channel_range(n_channels, n_devices, g, &s.first, &s.count);
    const int rc = hrfd_rx_create(s.count, s.device, &s.rx);
    if (rc != HRFD_OK)
    {
      fanout_free(f);
      return rc;
    }
  }
  // peer access between every pair of distinct devices (a copy between peers then goes over xGMI directly)
  for (uint32_t a = 0; a < n_devices; a++)
  {
    for (uint32_t b = 0; b < n_devices; b++)
    {
      const int da = f->shards[a].device, db = f->shards[b].device;
      if (da == db)
      {
        continue;
      }
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, da, db) == hipSuccess && can)
      {
        (void)hipSetDevice(da);
        const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
        if (e != hipSuccess)
        {
          (void)hipGetLastError();                         // (already enabled: fine)
        }
      }
    }
  }
  *out = f;
  return HRFD_OK;
}

extern "C" int hrfd_fanout_destroy(hrfd_fanout *f) { return fanout_free(f); }

extern "C" int hrfd_fanout_shards(hrfd_fanout *f, uint32_t *n_shards)
{
  if (f == nullptr || n_shards == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_fanout_shards: NULL");
  }
  *n_shards = (uint32_t)f->shards.size();
  return HRFD_OK;
}

// which shard owns a channel of the whole bank, and its index inside that shard
static bool fanout_locate(hrfd_fanout *f, uint32_t channel, uint32_t &shard, uint32_t &local)
{
  for (uint32_t g = 0; g < f->shards.size(); g++)
  {
    const hrfd_fanout::Shard &s = f->shards[g];
    if (channel >= s.first && channel < s.first + s.count)
    {
      shard = g;
      local = channel - s.first;
      return true;
    }
  }
  return false;
}

template <typename F>
static int fanout_for(hrfd_fanout *f, uint32_t channel, F fn)
{
  if (f == nullptr)
  {
    return fail(HRFD_EINVAL, "NULL handle");
  }
  if (channel == HRFD_ALL_CHANNELS)
  {
    for (hrfd_fanout::Shard &s : f->shards)
    {
      const int rc = fn(s.rx, HRFD_ALL_CHANNELS);
      if (rc != HRFD_OK) return rc;
    }
    return HRFD_OK;
  }
  uint32_t g = 0, local = 0;
  if (!fanout_locate(f, channel, g, local))
  {
    return fail(HRFD_EINVAL, "channel %u out of range (%u channels)", channel, f->n_channels);
  }
  return fn(f->shards[g].rx, local);
}

// the setters of hrfd_rx with channel numbers of the whole bank
extern "C" int hrfd_fanout_set_mode(hrfd_fanout *f, uint32_t channel, int mode)
{
  return fanout_for(f, channel, [&](hrfd_rx *rx, uint32_t c) { return hrfd_rx_set_mode(rx, c, mode); });
}
extern "C" int hrfd_fanout_set_gain(hrfd_fanout *f, uint32_t channel, int mode, float gain)
{
  return fanout_for(f, channel, [&](hrfd_rx *rx, uint32_t c) { return hrfd_rx_set_gain(rx, c, mode, gain); });
}
extern "C" int hrfd_fanout_set_threshold(hrfd_fanout *f, uint32_t channel, int32_t threshold)
{
  return fanout_for(f, channel, [&](hrfd_rx *rx, uint32_t c) { return hrfd_rx_set_threshold(rx, c, threshold); });
}

static int fanout_buffers(hrfd_fanout *f, uint32_t block_bytes, uint32_t n_blocks)
{
  if (block_bytes == 0 || (block_bytes & 1u) != 0 || block_bytes > HRFD_BLOCK_BYTES || n_blocks == 0)
  {
    return fail(HRFD_EINVAL, "block_bytes must be even, > 0 and <= %u, n_blocks > 0", HRFD_BLOCK_BYTES);
  }
  for (hrfd_fanout::Shard &s : f->shards)
  {
    HIP_TRY(hipSetDevice(s.device));
    const size_t units = (size_t)s.count * n_blocks;
    int rc;
    if (units * block_bytes > s.cap_iq || units * ((block_bytes + 511u) / 512u) * 2 > s.cap_pcm || units * 4 > s.cap_npcm)
    {
      HIP_TRY(hipStreamSynchronize(s.rx->stream));
    }
    if ((rc = grow((void **)&s.d_iq, &s.cap_iq, units * block_bytes)) != HRFD_OK) return rc;
    if ((rc = grow((void **)&s.d_pcm, &s.cap_pcm, units * ((block_bytes + 511u) / 512u) * sizeof(int16_t))) != HRFD_OK) return rc;
    if ((rc = grow((void **)&s.d_npcm, &s.cap_npcm, units * sizeof(uint32_t))) != HRFD_OK) return rc;
  }
  return HRFD_OK;
}

// d_iq_all [n_channels][n_blocks][block_bytes] on src_device -> every shard's input buffer: one peer copy per shard,
// each on the receiving shard's stream (they run at the same time, one xGMI link each).  src_stream: the stream of
// src_device on which d_iq_all was produced (its completion is awaited on the device, not by the host), or NULL when
// the data is complete already.
extern "C" int hrfd_fanout_scatter(hrfd_fanout *f, int src_device, const int8_t *d_iq_all, uint32_t block_bytes,
                                   uint32_t n_blocks, void *src_stream)
{
  if (f == nullptr || d_iq_all == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_fanout_scatter: NULL");
  }
  int rc = fanout_buffers(f, block_bytes, n_blocks);
  if (rc != HRFD_OK)
  {
    return rc;
  }
  const size_t per_channel = (size_t)n_blocks * block_bytes;
  hipEvent_t ready = nullptr;
  if (src_stream != nullptr)
  {
    HIP_TRY(hipSetDevice(src_device));
    for (hrfd_fanout::SrcEvent &e : f->src_events)
    {
      if (e.device == src_device)
      {
        ready = e.ev;
      }
    }
    if (ready == nullptr)
    {
      HIP_TRY(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
      f->src_events.push_back({src_device, ready});
    }
    HIP_TRY(hipEventRecord(ready, (hipStream_t)src_stream));
  }
  for (hrfd_fanout::Shard &s : f->shards)
  {
    HIP_TRY(hipSetDevice(s.device));
    if (ready != nullptr)
    {
      HIP_TRY(hipStreamWaitEvent(s.rx->stream, ready, 0));
    }
    const int8_t *src = d_iq_all + (size_t)s.first * per_channel;
    if (s.device == src_device)
    {
      HIP_TRY(hipMemcpyAsync(s.d_iq, src, (size_t)s.count * per_channel, hipMemcpyDeviceToDevice, s.rx->stream));
    }
    else
    {
      HIP_TRY(hipMemcpyPeerAsync(s.d_iq, s.device, src, src_device, (size_t)s.count * per_channel, s.rx->stream));
    }
  }
  f->block_bytes = block_bytes;
  f->n_blocks = n_blocks;
  return HRFD_OK;
}

// The shard's input buffer, for a host that feeds every device by itself (no scatter): [count][n_blocks][block_bytes].
extern "C" int hrfd_fanout_input(hrfd_fanout *f, uint32_t shard, uint32_t block_bytes, uint32_t n_blocks, int8_t **d_iq,
                                 uint32_t *first_channel, uint32_t *n_shard_channels)
{
  if (f == nullptr || shard >= f->shards.size() || d_iq == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_fanout_input: bad handle or shard");
  }
  const int rc = fanout_buffers(f, block_bytes, n_blocks);
  if (rc != HRFD_OK)
  {
    return rc;
  }
  f->block_bytes = block_bytes;
  f->n_blocks = n_blocks;
  *d_iq = f->shards[shard].d_iq;
  if (first_channel != nullptr) *first_channel = f->shards[shard].first;
  if (n_shard_channels != nullptr) *n_shard_channels = f->shards[shard].count;
  return HRFD_OK;
}

// IqDataProcessor::acceptIqData (IqDataProcessor.cc:926-1038) for every channel of every shard, n_blocks blocks each,
// from the shards' input buffers: asynchronous, every device at once.
extern "C" int hrfd_fanout_process(hrfd_fanout *f, uint32_t gain_db)
{
  if (f == nullptr || f->n_blocks == 0)
  {
    return fail(HRFD_ESTATE, "hrfd_fanout_process: nothing scattered yet");
  }
  for (hrfd_fanout::Shard &s : f->shards)
  {
    HIP_TRY(hipSetDevice(s.device));
    // mode NONE and squelched units produce no PCM: they read as zeros, not as the previous batch
    HIP_TRY(hipMemsetAsync(s.d_pcm, 0, (size_t)s.count * f->n_blocks * ((f->block_bytes + 511u) / 512u) * sizeof(int16_t), s.rx->stream));
    const int rc = hrfd_rx_process_device(s.rx, s.d_iq, (uint64_t)f->n_blocks * f->block_bytes, f->block_bytes, f->n_blocks,
                                          gain_db, s.d_pcm, s.d_npcm, nullptr, nullptr, nullptr, nullptr);
    if (rc != HRFD_OK)
    {
      return rc;
    }
  }
  f->gain_db = gain_db;
  f->in_flight = true;
  return HRFD_OK;
}

// Waits for every shard, replays the channels that failed their speculation on the exact path (hrfd_rx_process_block
// does the same), and gathers PCM [n_channels][n_blocks][block_bytes/512] and n_pcm [n_channels][n_blocks] (may be
// NULL) into buffers on dst_device: one peer copy per shard.  *n_replayed (may be NULL): channels that were replayed.
extern "C" int hrfd_fanout_collect(hrfd_fanout *f, int dst_device, int16_t *d_pcm_all, uint32_t *d_n_pcm_all,
                                   uint32_t *n_replayed)
{
  if (f == nullptr || !f->in_flight || d_pcm_all == nullptr)
  {
    return fail(HRFD_ESTATE, "hrfd_fanout_collect: no batch in flight, or no destination");
  }
  const uint32_t npcm = (f->block_bytes + 511u) / 512u;
  uint32_t replayed = 0;
  for (hrfd_fanout::Shard &s : f->shards)
  {
    HIP_TRY(hipSetDevice(s.device));
    uint32_t viol = 0;
    int rc = hrfd_rx_sync(s.rx, &viol);
    if (rc != HRFD_OK)
    {
      return rc;
    }
    if (viol != 0)
    {
      std::vector<uint32_t> redo;
      for (uint32_t c = 0; c < s.count; c++)
      {
        if (s.rx->h_fail[c] != 0) redo.push_back(c);
      }
      replayed += (uint32_t)redo.size();
      rc = rx_replay(s.rx, redo, s.d_iq, (uint64_t)f->n_blocks * f->block_bytes, f->block_bytes, f->n_blocks, f->gain_db,
                     s.d_pcm, s.d_npcm, nullptr, nullptr, nullptr, s.rx->stream, false);
      if (rc != HRFD_OK)
      {
        return rc;
      }
    }
    const size_t units = (size_t)s.count * f->n_blocks, off = (size_t)s.first * f->n_blocks;
    if (s.device == dst_device)
    {
      HIP_TRY(hipMemcpyAsync(d_pcm_all + off * npcm, s.d_pcm, units * npcm * sizeof(int16_t), hipMemcpyDeviceToDevice, s.rx->stream));
      if (d_n_pcm_all != nullptr)
      {
        HIP_TRY(hipMemcpyAsync(d_n_pcm_all + off, s.d_npcm, units * sizeof(uint32_t), hipMemcpyDeviceToDevice, s.rx->stream));
      }
    }
    else
    {
      HIP_TRY(hipMemcpyPeerAsync(d_pcm_all + off * npcm, dst_device, s.d_pcm, s.device, units * npcm * sizeof(int16_t), s.rx->stream));
      if (d_n_pcm_all != nullptr)
      {
        HIP_TRY(hipMemcpyPeerAsync(d_n_pcm_all + off, dst_device, s.d_npcm, s.device, units * sizeof(uint32_t), s.rx->stream));
      }
    }
  }
  for (hrfd_fanout::Shard &s : f->shards)
  {
    HIP_TRY(hipSetDevice(s.device));
    HIP_TRY(hipStreamSynchronize(s.rx->stream));
  }
  f->in_flight = false;
  if (n_replayed != nullptr)
  {
    *n_replayed = replayed;
  }
  return HRFD_OK;
}
