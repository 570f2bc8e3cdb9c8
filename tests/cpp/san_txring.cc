// Sanitizer harness for the host-only transmit ring (hackrfdiags_amd/csrc/hrfd_txring.hip), CPU only: the file is
// compiled here as plain C++ under -fsanitize=address,undefined and again under -fsanitize=thread
// (tests/test_sanitizers.py).  A writer thread (the PCM reader of BasebandDataProcessor.cc:869) and the transmit
// callback's reader (getNextFilledBuffer, :476-606) run against each other over three channels, paced the way the
// reference's two threads are (one block per 64 ms each, here microseconds), through a start, a stop and a restart.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "../../include/hrfd.h"
static int fail(int code, const char *, ...) { return code; }
#include "../../hackrfdiags_amd/csrc/hrfd_txring.hip"

int main()
{
  const uint32_t C = 3;
  hrfd_txring *r = nullptr;
  if (hrfd_txring_create(C, &r) != HRFD_OK || hrfd_txring_create(0, &r) == HRFD_OK || hrfd_txring_write(r, C, nullptr) == HRFD_OK)
  {
    return 2;
  }
  hrfd_txring_set_running(r, HRFD_ALL_CHANNELS, 1);
  const int N = 3000;
  std::atomic<int> written{0};
  std::thread writer([&] {
    std::vector<int16_t> blk(512);
    for (int i = 0; i < N; i++)
    {
      for (uint32_t c = 0; c < C; c++)
      {
        for (int k = 0; k < 512; k++)
        {
          blk[k] = (int16_t)(i * 7 + (int)c * 1000 + k);
        }
        hrfd_txring_write(r, c, blk.data());
      }
      written.store(i + 1, std::memory_order_release);
      if ((i % 3) == 0)
      {
        std::this_thread::sleep_for(std::chrono::microseconds(30));
      }
    }
  });
  long long sum = 0;
  std::vector<int16_t> batch(C * 512);
  int reads = 0;
  while (written.load(std::memory_order_acquire) < N)
  {
    // the reader keeps its distance the way the transmit callback does: it only runs when the writer is ahead
    if (written.load(std::memory_order_acquire) > reads + 7)
    {
      hrfd_txring_read_batch(r, batch.data());
      reads++;
      for (int16_t v : batch)
      {
        sum += v;
      }
    }
    else
    {
      std::this_thread::sleep_for(std::chrono::microseconds(5));
    }
  }
  writer.join();
  uint32_t st[6];
  for (uint32_t c = 0; c < C; c++)
  {
    if (hrfd_txring_stats(r, c, st) != HRFD_OK || st[0] != (uint32_t)N || st[1] == 0 || st[4] >= 16 || st[5] >= 16)
    {
      return 3;
    }
  }
  hrfd_txring_set_running(r, 1, 0);                        // stop one channel: silence, no advance
  hrfd_txring_read_batch(r, batch.data());
  for (int k = 0; k < 512; k++)
  {
    if (batch[512 + k] != 0)
    {
      return 4;
    }
  }
  hrfd_txring_set_running(r, 1, 1);                        // restart: re-synchronises half a ring behind the writer
  hrfd_txring_read_batch(r, batch.data());
  hrfd_txring_destroy(r);
  printf("san_txring ok: %d reads, checksum %lld\n", reads, sum);
  return 0;
}
