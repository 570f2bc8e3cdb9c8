// Which shape of a pure store stream reaches the chip's write rate?  k_mod writes 32 KiB per workgroup (one tile of
// one channel: eight rounds of 256 lanes x 16 bytes); a fill of the same 4 GiB is the yardstick.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o store_shapes store_shapes.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

// A: one workgroup of T threads per chunk of R rounds x T x 16 bytes
template <int T, int R, bool NT>
__global__ __launch_bounds__(T) void k_chunks(uint4 *out, size_t chunks)
{
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  for (size_t ch = blockIdx.x; ch < chunks; ch += gridDim.x)
  {
    uint4 *o = out + ch * (size_t)(R * T) + threadIdx.x;
#pragma unroll
    for (int r = 0; r < R; r++)
    {
      const uint4 v = make_uint4((uint32_t)ch, (uint32_t)r, threadIdx.x, 7u);
      if (NT) __builtin_nontemporal_store(u4{v.x, v.y, v.z, v.w}, reinterpret_cast<u4 *>(o + r * T));
      else o[r * T] = v;
    }
  }
}

// the same chunks dealt so that every XCD (workgroup ids go round the eight XCDs) writes a contiguous eighth of the buffer,
// or runs of `run` consecutive chunks
template <int T, int R>
__global__ __launch_bounds__(T) void k_chunks_xcd(uint4 *out, size_t chunks, uint32_t run)
{
  const size_t per = chunks / 8;
  const uint32_t x = blockIdx.x & 7u, i = blockIdx.x >> 3;
  size_t ch;
  if (run == 0)
  {
    ch = (size_t)x * per + i;                            // XCD x: chunks [x per, (x + 1) per)
  }
  else
  {
    ch = ((size_t)(i / run) * 8 + x) * run + (i % run);  // XCD x: runs of `run` consecutive chunks, the runs go round the XCDs
  }
  uint4 *o = out + ch * (size_t)(R * T) + threadIdx.x;
#pragma unroll
  for (int r = 0; r < R; r++)
  {
    o[r * T] = make_uint4((uint32_t)ch, (uint32_t)r, threadIdx.x, 7u);
  }
}

template <typename K>
static void run(const char *name, K launch, size_t bytes)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f, sum = 0;
  for (int rep = 0; rep < 60; rep++)
  {
    hipEventRecord(e0, 0);
    launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 20) { sum += ms; if (ms < best) best = ms; }
  }
  printf("%-64s mean %.4f ms  min %.4f ms  %.2f TB/s\n", name, sum / 40, best, bytes / (sum / 40 * 1e-3) / 1e12);
}

int main()
{
  const size_t bytes = (size_t)1024 * 16 * 131072 * 2;   // 1024 channels x 16 blocks x 131072 IQ pairs x 2 bytes = 4 GiB
  uint4 *out;
  hipMalloc(&out, bytes);
  const size_t chunks32k = bytes / 32768;
  run("fill (hipMemsetAsync)", [&] { hipMemsetAsync(out, 1, bytes, 0); }, bytes);
  run("256 threads x 8 rounds, one workgroup per 32 KiB chunk", [&] { hipLaunchKernelGGL((k_chunks<256, 8, false>), dim3((uint32_t)chunks32k), dim3(256), 0, 0, out, chunks32k); }, bytes);
  run("  the same, nontemporal", [&] { hipLaunchKernelGGL((k_chunks<256, 8, true>), dim3((uint32_t)chunks32k), dim3(256), 0, 0, out, chunks32k); }, bytes);
  run("256 x 8, one workgroup per 32 KiB chunk, every XCD a contiguous eighth", [&] { hipLaunchKernelGGL((k_chunks_xcd<256, 8>), dim3((uint32_t)chunks32k), dim3(256), 0, 0, out, chunks32k, 0u); }, bytes);
  for (uint32_t rl : {2u, 4u, 8u, 16u, 128u})
  {
    char nm[96];
    snprintf(nm, sizeof nm, "256 x 8, 32 KiB chunks, an XCD takes runs of %u consecutive chunks", rl);
    run(nm, [&] { hipLaunchKernelGGL((k_chunks_xcd<256, 8>), dim3((uint32_t)chunks32k), dim3(256), 0, 0, out, chunks32k, rl); }, bytes);
  }
  run("256 threads x 8 rounds, 2048 persistent workgroups", [&] { hipLaunchKernelGGL((k_chunks<256, 8, false>), dim3(2048), dim3(256), 0, 0, out, chunks32k); }, bytes);
  run("256 threads x 8 rounds, 8192 persistent workgroups", [&] { hipLaunchKernelGGL((k_chunks<256, 8, false>), dim3(8192), dim3(256), 0, 0, out, chunks32k); }, bytes);
  run("  the same, nontemporal", [&] { hipLaunchKernelGGL((k_chunks<256, 8, true>), dim3(8192), dim3(256), 0, 0, out, chunks32k); }, bytes);
  run("1024 threads x 8 rounds, one workgroup per 128 KiB chunk", [&] { hipLaunchKernelGGL((k_chunks<1024, 8, false>), dim3((uint32_t)(bytes / 131072)), dim3(1024), 0, 0, out, bytes / 131072); }, bytes);
  run("256 threads x 32 rounds, one workgroup per 128 KiB chunk", [&] { hipLaunchKernelGGL((k_chunks<256, 32, false>), dim3((uint32_t)(bytes / 131072)), dim3(256), 0, 0, out, bytes / 131072); }, bytes);
  run("64 threads x 8 rounds, one workgroup per 8 KiB chunk", [&] { hipLaunchKernelGGL((k_chunks<64, 8, false>), dim3((uint32_t)(bytes / 8192)), dim3(64), 0, 0, out, bytes / 8192); }, bytes);
  run("256 threads x 1 round, one workgroup per 4 KiB", [&] { hipLaunchKernelGGL((k_chunks<256, 1, false>), dim3((uint32_t)(bytes / 4096)), dim3(256), 0, 0, out, bytes / 4096); }, bytes);
  return 0;
}
