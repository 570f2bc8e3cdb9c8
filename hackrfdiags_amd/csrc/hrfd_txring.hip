// hackrfdiags_amd/csrc/hrfd_txring.hip -- the transmit side's PCM ring for many channels
// (second half).  Host code only.
//
// BasebandDataProcessor keeps a 16-slot ring of 512-sample PCM blocks between the thread that
// reads PCM and the transmit callback, and paces it (BasebandDataProcessor.cc:476-606): when the
// writer runs more than 10 blocks ahead a block is dropped, when it is less than 6 ahead the
// previous block is sent again; the first read after start() jumps to a table entry half a ring
// behind the writer; while the stream is not running the reader gets zeros.  hrfd_txring keeps
// one such ring per channel, with exactly that policy, and gathers one block per channel into
// the batch buffer that hrfd_mod_process() takes.
#include <stdint.h>
#include <string.h>

#include <mutex>
#include <vector>

struct hrfd_txring
{
  static constexpr int kRing = 16, kBlock = 512;          // PCM_RING_SIZE, PCM_BLOCK_SIZE
  struct Chan
  {
    uint32_t writer = kRing - 1, reader = 8;              // ctor :78-79 (table[15] = 7 ... see create)
    bool running = false, synchronized = false;
    uint32_t produced = 0, consumed = 0, dropped = 0, added = 0;
    std::mutex writer_lock;                               // the reference's writerLock
    int16_t buf[kRing][kBlock];
  };
  std::vector<Chan> ch;
};

namespace {
const int kReaderStart[16] = {8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7};   // :19-20
}

extern "C" int hrfd_txring_create(uint32_t n_channels, hrfd_txring **out)
{
  if (out == nullptr || n_channels == 0)
  {
    return fail(HRFD_EINVAL, "hrfd_txring_create: need n_channels > 0 and a result pointer");
  }
  hrfd_txring *r = new hrfd_txring;
  r->ch = std::vector<hrfd_txring::Chan>(n_channels);
  for (auto &c : r->ch)
  {
    c.writer = hrfd_txring::kRing - 1;
    c.reader = (uint32_t)kReaderStart[c.writer];
    memset(c.buf, 0, sizeof(c.buf));
  }
  *out = r;
  return HRFD_OK;
}

extern "C" int hrfd_txring_destroy(hrfd_txring *r)
{
  delete r;
  return HRFD_OK;
}

// BasebandDataProcessor::start() / stop() as far as the ring is concerned (:306-356)
extern "C" int hrfd_txring_set_running(hrfd_txring *r, uint32_t channel, int running)
{
  if (r == nullptr || (channel != HRFD_ALL_CHANNELS && channel >= r->ch.size()))
  {
    return fail(HRFD_EINVAL, "hrfd_txring_set_running: bad handle or channel");
  }
  for (uint32_t c = 0; c < r->ch.size(); c++)
  {
    if (channel != HRFD_ALL_CHANNELS && channel != c)
    {
      continue;
    }
    hrfd_txring::Chan &k = r->ch[c];
    if (running)
    {
      k.running = true;
    }
    else if (k.running)
    {
      k.running = false;
      k.synchronized = false;
    }
  }
  return HRFD_OK;
}

// getNextUnfilledBuffer (:410-425) + the copy the reader thread does into it (:869)
extern "C" int hrfd_txring_write(hrfd_txring *r, uint32_t channel, const int16_t *pcm512)
{
  if (r == nullptr || pcm512 == nullptr || channel >= r->ch.size())
  {
    return fail(HRFD_EINVAL, "hrfd_txring_write: bad handle, channel or buffer");
  }
  hrfd_txring::Chan &k = r->ch[channel];
  uint32_t w;
  {
    std::lock_guard<std::mutex> g(k.writer_lock);
    k.writer++;
    k.writer %= hrfd_txring::kRing;
    w = k.writer;
  }
  memcpy(k.buf[w], pcm512, sizeof(k.buf[w]));
  k.produced++;
  return HRFD_OK;
}

// getNextFilledBuffer (:476-606) for every channel: batch [n_channels][512], the input of
// hrfd_mod_process(h, batch, 512, ...)
extern "C" int hrfd_txring_read_batch(hrfd_txring *r, int16_t *batch)
{
  if (r == nullptr || batch == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_txring_read_batch: NULL");
  }
  for (size_t c = 0; c < r->ch.size(); c++)
  {
    hrfd_txring::Chan &k = r->ch[c];
    int16_t *dst = batch + c * hrfd_txring::kBlock;
    int32_t u;
    {
      std::lock_guard<std::mutex> g(k.writer_lock);
      u = (int32_t)k.writer;
    }
    const int32_t l = (int32_t)k.reader;
    if (u < l)
    {
      u += hrfd_txring::kRing - 1;                        // as written in the reference (:513)
    }
    const int32_t lag = u - l;
    if (lag > 10)
    {
      k.reader = (k.reader + 1) % hrfd_txring::kRing;     // writer too far ahead: drop a block
      k.dropped++;
    }
    else if (lag < 6)
    {
      int32_t d = (int32_t)k.reader - 1;                  // writer too close: send the previous block again
      if (d < 0)
      {
        d += hrfd_txring::kRing;
      }
      k.reader = (uint32_t)d;
      k.added++;
    }
    if (k.running)
    {
      if (!k.synchronized)
      {
        k.synchronized = true;
        std::lock_guard<std::mutex> g(k.writer_lock);
        k.reader = (uint32_t)kReaderStart[k.writer];
      }
      memcpy(dst, k.buf[k.reader], sizeof(k.buf[0]));
      k.reader = (k.reader + 1) % hrfd_txring::kRing;
      k.consumed++;
    }
    else
    {
      memset(dst, 0, sizeof(k.buf[0]));                   // zeroPcmBuffer
    }
  }
  return HRFD_OK;
}

// {buffersProduced, buffersConsumed, pcmBlocksDropped, pcmBlocksAdded, pcmWriterIndex, pcmReaderIndex}
extern "C" int hrfd_txring_stats(hrfd_txring *r, uint32_t channel, uint32_t *out6)
{
  if (r == nullptr || out6 == nullptr || channel >= r->ch.size())
  {
    return fail(HRFD_EINVAL, "hrfd_txring_stats: bad handle, channel or buffer");
  }
  const hrfd_txring::Chan &k = r->ch[channel];
  out6[0] = k.produced; out6[1] = k.consumed; out6[2] = k.dropped; out6[3] = k.added;
  out6[4] = k.writer; out6[5] = k.reader;
  return HRFD_OK;
}
