// What does the flow kernel's ACCESS PATTERN reach when nothing is computed?  256 (or 1024) persistent workgroups, one
// contiguous 4 MiB "channel" each, 12 waves per workgroup taking 8 KiB units from an LDS counter in order, eight 16-byte
// loads per lane and unit (lane-contiguous 16 bytes: 1 KiB per wave instruction), the next unit requested before the
// current one is consumed -- against the plain shape of hrfd_membw.hip (32 KiB per workgroup, 16 bytes per lane, every XCD
// one contiguous eighth of the buffer) in the same run.  Variants: channel -> workgroup mapping (channels of an XCD
// interleaved, as k_rx_wbfm_flow has them, or contiguous), unit size, lead.
// Build: hipcc --offload-arch=gfx950 -O3 -o flow_read flow_read.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

constexpr int kBwThreads = 256, kBwRounds = 8;

__global__ __launch_bounds__(kBwThreads) void k_plain(const uint4 *__restrict__ in, size_t chunks, uint32_t *sink)
{
  const size_t per = chunks / 8;
  const uint32_t x = blockIdx.x & 7u, i = blockIdx.x >> 3;
  const uint4 *p = in + ((size_t)x * per + i) * (size_t)(kBwRounds * kBwThreads) + threadIdx.x;
  uint4 q[kBwRounds];
#pragma unroll
  for (int r = 0; r < kBwRounds; r++) q[r] = p[r * kBwThreads];
  uint32_t acc = 0;
#pragma unroll
  for (int r = 0; r < kBwRounds; r++) acc ^= q[r].x ^ q[r].y ^ q[r].z ^ q[r].w;
  if (acc == 0x9e3779b9u) sink[0] = acc;
}

// MAP 0: channel = blockIdx (an XCD gets channels x, x + 8, ...); 1: an XCD's channels are contiguous in memory
// LEAD: units requested ahead of the one being consumed (1 or 2);  UNIT16: 16-byte loads per lane and unit (8 = 8 KiB units)
template <int MAP, int LEAD, int UNIT16>
__global__ __launch_bounds__(1024) void k_flow(const uint4 *__restrict__ in, const uint32_t units_per_chan, const int stream_waves, uint32_t *sink)
{
  __shared__ uint32_t next;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) next = 0;
  __syncthreads();
  if (wave >= stream_waves) return;
  const uint32_t nb = gridDim.x;
  const uint32_t chan = (MAP == 0) ? blockIdx.x : (blockIdx.x & 7u) * (nb / 8) + (blockIdx.x >> 3);
  const uint4 *base = in + (size_t)chan * units_per_chan * (UNIT16 * 64) + lane;
  auto grab = [&]() -> uint32_t {
    uint32_t v = 0;
    if (lane == 0) v = atomicAdd(&next, 1u);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
  };
  uint32_t acc = 0;
  uint4 q[LEAD + 1][UNIT16];
  uint32_t u[LEAD + 1];
#pragma unroll
  for (int l = 0; l < LEAD; l++)
  {
    u[l] = grab();
    if (u[l] < units_per_chan)
    {
#pragma unroll
      for (int r = 0; r < UNIT16; r++) q[l][r] = base[(size_t)u[l] * (UNIT16 * 64) + r * 64];
    }
  }
  while (u[0] < units_per_chan)
  {
    u[LEAD] = grab();
    if (u[LEAD] < units_per_chan)
    {
#pragma unroll
      for (int r = 0; r < UNIT16; r++) q[LEAD][r] = base[(size_t)u[LEAD] * (UNIT16 * 64) + r * 64];
    }
#pragma unroll
    for (int r = 0; r < UNIT16; r++) acc ^= q[0][r].x ^ q[0][r].y ^ q[0][r].z ^ q[0][r].w;
#pragma unroll
    for (int l = 0; l < LEAD; l++)
    {
      u[l] = u[l + 1];
#pragma unroll
      for (int r = 0; r < UNIT16; r++) q[l][r] = q[l + 1][r];
    }
  }
  if (acc == 0x9e3779b9u) sink[0] = acc;
}

template <class F>
static double time_ms(F &&launch, hipStream_t s, int warm, int reps)
{
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < warm; i++) launch();
  std::vector<float> t;
  for (int k = 0; k < 5; k++)
  {
    hipEventRecord(a, s);
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    t.push_back(ms / reps);
  }
  std::sort(t.begin(), t.end());
  return t[2];
}

int main()
{
  const size_t bytes = 1ull << 30;                         // the headline's launch: 256 channels x 4 MiB
  uint4 *buf; uint32_t *sink;
  hipMalloc(&buf, bytes); hipMalloc(&sink, 64);
  hipMemset(buf, 1, bytes); hipMemset(sink, 0, 64);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const size_t chunks = bytes / 32768;
  auto report = [&](const char *name, double ms) { printf("%-64s %.4f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
  report("plain: 32 KiB per workgroup, XCD-contiguous eighths", time_ms([&] { hipLaunchKernelGGL(k_plain, dim3((uint32_t)chunks), dim3(kBwThreads), 0, s, buf, chunks, sink); }, s, 50, 50));
  for (int nchan : {256, 1024})
  {
    const uint32_t upc8 = (uint32_t)(bytes / nchan / 8192), upc4 = (uint32_t)(bytes / nchan / 4096);
    char nm[160];
    for (int sw : {12, 16, 8})
    {
      snprintf(nm, sizeof nm, "flow: %d channels, %d waves, 8 KiB units, lead 1, interleaved", nchan, sw);
      report(nm, time_ms([&] { hipLaunchKernelGGL((k_flow<0, 1, 8>), dim3(nchan), dim3(1024), 0, s, buf, upc8, sw, sink); }, s, 50, 50));
      snprintf(nm, sizeof nm, "flow: %d channels, %d waves, 8 KiB units, lead 2, interleaved", nchan, sw);
      report(nm, time_ms([&] { hipLaunchKernelGGL((k_flow<0, 2, 8>), dim3(nchan), dim3(1024), 0, s, buf, upc8, sw, sink); }, s, 50, 50));
      snprintf(nm, sizeof nm, "flow: %d channels, %d waves, 8 KiB units, lead 1, XCD-contiguous", nchan, sw);
      report(nm, time_ms([&] { hipLaunchKernelGGL((k_flow<1, 1, 8>), dim3(nchan), dim3(1024), 0, s, buf, upc8, sw, sink); }, s, 50, 50));
      snprintf(nm, sizeof nm, "flow: %d channels, %d waves, 4 KiB units, lead 2, interleaved", nchan, sw);
      report(nm, time_ms([&] { hipLaunchKernelGGL((k_flow<0, 2, 4>), dim3(nchan), dim3(1024), 0, s, buf, upc4, sw, sink); }, s, 50, 50));
    }
  }
  report("plain again", time_ms([&] { hipLaunchKernelGGL(k_plain, dim3((uint32_t)chunks), dim3(kBwThreads), 0, s, buf, chunks, sink); }, s, 50, 50));
  return 0;
}
