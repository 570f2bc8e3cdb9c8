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
    uint32_t writer = kRing - 1, reader = 7;              // ctor :78-79: the last slot, and the start slot that goes with it
    bool running = false, synchronized = false;
    uint32_t produced = 0, consumed = 0, dropped = 0, added = 0;
    std::mutex writer_lock;                               // the reference's writerLock
    int16_t buf[kRing][kBlock];
  };
  std::vector<Chan> ch;
};

namespace {
// where a reader starts for a given writer slot: half a ring behind it (the reference's start table, :19-20)
inline uint32_t reader_start(uint32_t writer) { return (writer + hrfd_txring::kRing / 2) % hrfd_txring::kRing; }
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
    c.reader = reader_start(c.writer);
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

// One read of a channel's ring, the policy of getNextFilledBuffer (BasebandDataProcessor.cc:476-606) restated:
//
//   lag     how many blocks the writer is ahead of the reader.  The reference unwraps a writer index that is
//           numerically below the reader by adding RING - 1 (not RING), so a wrapped lag reads one short -- the
//           pacing decisions below depend on it and the oracle tests pin it.
//   nudge   the pacing decision as a signed reader adjustment: +1 skips a block (lag above 10: the writer is
//           running away), -1 steps back so that the previous block goes out again (lag below 6: the writer is
//           about to be caught), 0 otherwise.  It is applied whether or not the stream is running, as in the
//           reference, and counted as a dropped / an added block.
//   start   the first read after start() ignores where the reader was and takes the slot half a ring behind
//           the writer (the reference's start table is (writer + 8) mod 16).
//   idle    a stream that is not running hands out silence and does not advance.
namespace {
constexpr uint32_t kSlots = hrfd_txring::kRing;

inline uint32_t ring_add(uint32_t index, int delta) { return (index + kSlots + (uint32_t)delta) % kSlots; }

inline int writer_lag(uint32_t writer, uint32_t reader)
{
  return (writer >= reader) ? (int)(writer - reader) : (int)(writer + (kSlots - 1) - reader);
}

void read_one(hrfd_txring::Chan &k, int16_t *dst)
{
  uint32_t writer_now;
  {
    std::lock_guard<std::mutex> g(k.writer_lock);
    writer_now = k.writer;
  }
  const int lag = writer_lag(writer_now, k.reader);
  const int nudge = (lag > 10) - (lag < 6);
  k.reader = ring_add(k.reader, nudge);
  k.dropped += (nudge > 0);
  k.added += (nudge < 0);
  if (!k.running)
  {
    memset(dst, 0, sizeof(k.buf[0]));
    return;
  }
  if (!k.synchronized)
  {
    std::lock_guard<std::mutex> g(k.writer_lock);
    k.reader = reader_start(k.writer);
    k.synchronized = true;
  }
  memcpy(dst, k.buf[k.reader], sizeof(k.buf[0]));
  k.reader = ring_add(k.reader, 1);
  k.consumed++;
}
} // namespace

// batch [n_channels][512]: one block per channel, the input of hrfd_mod_process(h, batch, 512, ...)
extern "C" int hrfd_txring_read_batch(hrfd_txring *r, int16_t *batch)
{
  if (r == nullptr || batch == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_txring_read_batch: NULL");
  }
  for (size_t c = 0; c < r->ch.size(); c++)
  {
    read_one(r->ch[c], batch + c * hrfd_txring::kBlock);
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
