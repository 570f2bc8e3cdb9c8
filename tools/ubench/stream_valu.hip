// Microbenchmark: can VALU work hide under an HBM-bound stream on gfx950?
// Every wave streams a contiguous run of 1 KiB chunks (buffer-like global dwordx4
// loads, DEPTH in flight) and performs NV packed-math instructions per chunk on
// the loaded data.  Reports wall time for NV = 0..; perfect overlap keeps the
// time flat until the VALU time exceeds the stream time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

template <int NV, int DEPTH, bool E32, int GATHER>
__global__ __launch_bounds__(1024, 8) void k_stream(const uint4 *__restrict__ in, uint32_t *out, int chunks_per_wave, const uint32_t *__restrict__ lut)
{
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_readcyclecounter();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t base = ((size_t)blockIdx.x * 16 + wave) * (size_t)chunks_per_wave * 64;
  const uint4 *p = in + base + lane;
  uint4 q[DEPTH];
#pragma unroll
  for (int k = 0; k < DEPTH; k++) q[k] = p[(size_t)k * 64];
  uint32_t acc = 0;
  int c = 0;
  for (; c + DEPTH <= chunks_per_wave; c += DEPTH)
  {
#pragma unroll
    for (int k = 0; k < DEPTH; k++)
    {
      uint32_t a = q[k].x, b = q[k].y, d = q[k].z, e = q[k].w;
      const int nxt = min(c + k + DEPTH, chunks_per_wave - 1);
      q[k] = p[(size_t)nxt * 64];
#pragma unroll
      for (int i = 0; i < NV / 4; i++)
      {
        if (E32)
          asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(d), "+v"(e));
        else
          asm volatile("v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %1, %1, %2\n v_pk_add_u16 %2, %2, %3\n v_pk_add_u16 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(d), "+v"(e));
      }
      if (GATHER == 1)        // divergent: index from the data (uniformly random over 64K entries)
      {
        acc += lut[(a ^ (b >> 3) ^ (d << 2) ^ e) & 0xffffu];
      }
      else if (GATHER == 2)   // concentrated: gaussian-like index around the table centre (sum of 4 bytes)
      {
        const uint32_t i = ((a & 0xff) + ((a >> 8) & 0xff) + ((a >> 16) & 0xff) + (a >> 24)) >> 2;
        const uint32_t j = ((b & 0xff) + ((b >> 8) & 0xff) + ((b >> 16) & 0xff) + (b >> 24)) >> 2;
        acc += lut[(j << 8) | i];
      }
      else if (GATHER == 3)   // coalesced
      {
        acc += lut[((a & 0xff) << 8) | lane];
      }
      acc += a ^ b ^ d ^ e;
    }
  }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
  if (threadIdx.x == 0)
  {
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    out[4096 + 2 * blockIdx.x] = (uint32_t)(t1 - t0);
    out[4096 + 2 * blockIdx.x + 1] = (uint32_t)(r1 - r0);
  }
}


// Same stream, but every lane owns 64 CONTIGUOUS bytes of each 4 KiB piece (four dwordx4 loads whose
// lanes are 64 bytes apart): what a "four consecutive groups per lane" layout of the WBFM chunk loop
// would issue.  Does the per-CU L1 keep up with 64-byte-strided 16-byte requests?
template <int NV, int DEPTH>
__global__ __launch_bounds__(1024, 8) void k_stream_strided(const uint4 *__restrict__ in, uint32_t *out, int chunks_per_wave)
{
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_readcyclecounter();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t base = ((size_t)blockIdx.x * 16 + wave) * (size_t)chunks_per_wave * 64;
  const uint4 *p = in + base + lane * 4;                  // lane stride 64 bytes
  const int pieces = chunks_per_wave / 4;                 // 4 KiB pieces
  uint4 q[DEPTH][4];
#pragma unroll
  for (int k = 0; k < DEPTH; k++)
#pragma unroll
    for (int j = 0; j < 4; j++) q[k][j] = p[(size_t)k * 256 + j];
  uint32_t acc = 0;
  for (int c = 0; c + DEPTH <= pieces; c += DEPTH)
  {
#pragma unroll
    for (int k = 0; k < DEPTH; k++)
    {
      const int nxt = min(c + k + DEPTH, pieces - 1);
#pragma unroll
      for (int j = 0; j < 4; j++)
      {
        uint32_t a = q[k][j].x, b = q[k][j].y, d = q[k][j].z, e = q[k][j].w;
        q[k][j] = p[(size_t)nxt * 256 + j];
#pragma unroll
        for (int i = 0; i < NV / 4; i++)
          asm volatile("v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %1, %1, %2\n v_pk_add_u16 %2, %2, %3\n v_pk_add_u16 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(d), "+v"(e));
        acc += a ^ b ^ d ^ e;
      }
    }
  }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
  if (threadIdx.x == 0)
  {
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    out[4096 + 2 * blockIdx.x] = (uint32_t)(t1 - t0);
    out[4096 + 2 * blockIdx.x + 1] = (uint32_t)(r1 - r0);
  }
}

template <int NV, int DEPTH>
static void run_strided(const uint4 *in, uint32_t *out, int grid, int cpw, double bytes)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < 5; r++)
  {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_stream_strided<NV, DEPTH>), dim3(grid), dim3(1024), 0, 0, in, out, cpw);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  printf("strided 64 B per lane, NV %3d depth %d x 4 KiB: %.4f ms  %.0f GB/s\n", NV, DEPTH, best, bytes / best / 1e6);
}

template <int NV, int DEPTH, bool E32, int GATHER>
static void run(const uint4 *in, uint32_t *out, int grid, int cpw, double bytes, const uint32_t *lut)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < 5; r++)
  {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_stream<NV, DEPTH, E32, GATHER>), dim3(grid), dim3(1024), 0, 0, in, out, cpw, lut);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  static uint32_t h[3 * 4096];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  double ts = 0, rs = 0;
  for (int i = 0; i < grid; i++) { ts += h[4096 + 2 * i]; rs += h[4096 + 2 * i + 1]; }
  const double instr_per_simd = (double)grid * 16 / 1024.0 * cpw * NV;
  printf("gather %d NV %3d depth %d %s: %.4f ms  %.0f GB/s   valu-only estimate %.4f ms  shader clock %.0f MHz\n", GATHER, NV, DEPTH, E32 ? "e32 " : "vop3", best, bytes / best / 1e6,
         instr_per_simd * (E32 ? 1.05e-6 : 1.8e-6), ts / rs * 100.0);
}

int main()
{
  const int grid = 4096, cpw = 16;                      // 4096 WGs x 16 waves x 16 chunks x 1 KiB = 1 GiB
  const size_t bytes = (size_t)grid * 16 * cpw * 1024;
  uint4 *in; uint32_t *out;
  hipMalloc(&in, bytes); hipMalloc(&out, 3 * 4096 * 4);
  {
    // pseudo-random bytes so that the gather indices spread
    uint32_t *h = (uint32_t *)malloc(bytes);
    uint32_t x = 12345;
    for (size_t i = 0; i < bytes / 4; i++) { x = x * 1664525u + 1013904223u; h[i] = x ^ (x >> 13); }
    hipMemcpy(in, h, bytes, hipMemcpyHostToDevice);
    free(h);
  }
  uint32_t *lut; hipMalloc(&lut, 65536 * 4); hipMemset(lut, 0, 65536 * 4);
  run<0, 4, false, 0>(in, out, grid, cpw, bytes, lut);
  run<64, 4, false, 0>(in, out, grid, cpw, bytes, lut);
  run<96, 4, false, 0>(in, out, grid, cpw, bytes, lut);
  run<128, 4, false, 0>(in, out, grid, cpw, bytes, lut);
  run<192, 4, false, 0>(in, out, grid, cpw, bytes, lut);
  run<192, 4, true, 0>(in, out, grid, cpw, bytes, lut);
  run_strided<0, 1>(in, out, grid, cpw, bytes);
  run_strided<0, 2>(in, out, grid, cpw, bytes);
  run_strided<64, 1>(in, out, grid, cpw, bytes);
  run_strided<96, 1>(in, out, grid, cpw, bytes);
  run_strided<96, 2>(in, out, grid, cpw, bytes);
  run<0, 4, false, 1>(in, out, grid, cpw, bytes, lut);
  run<64, 4, false, 1>(in, out, grid, cpw, bytes, lut);
  run<96, 4, false, 1>(in, out, grid, cpw, bytes, lut);
  run<0, 4, false, 2>(in, out, grid, cpw, bytes, lut);
  run<64, 4, false, 2>(in, out, grid, cpw, bytes, lut);
  run<96, 4, false, 2>(in, out, grid, cpw, bytes, lut);
  run<0, 4, false, 3>(in, out, grid, cpw, bytes, lut);
  run<64, 4, false, 3>(in, out, grid, cpw, bytes, lut);
  run<96, 4, false, 3>(in, out, grid, cpw, bytes, lut);
  return 0;
}
