// Microbenchmark: cost per step of the de-emphasis recurrence y = v - a1*y (two dependent
// VOP2 instructions) as one wave runs it out of LDS, alone on its CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

template <int MODE>   // 0: chain only (inputs in registers), 1: + 64-bit LDS reads, 2: + reads and writes
__global__ __launch_bounds__(1024) void k(unsigned long long *out, int steps, int stride, int other_waves_busy)
{
  __shared__ __attribute__((aligned(16))) uint32_t lds[18048];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 18048; i += blockDim.x) lds[i] = f2u(1.0f + (i & 255) * 0.001f);
  __syncthreads();
  if (wave != 0)
  {
    if (other_waves_busy)
    {
      // keep the SIMDs busy with packed math (like a neighbouring workgroup in phase A)
      unsigned a = threadIdx.x, b = 0x00010002;
      for (int i = 0; i < steps * 2; i++)
        asm volatile("v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %0, %0, %1" : "+v"(a) : "v"(b));
      if (a == 0x12345678) out[1] = a;
    }
    return;
  }
  __builtin_amdgcn_s_setprio(3);
  const uint32_t *in = lds + lane * stride;
  uint32_t *o = lds + lane * stride;
  float y = 0.5f;
  const float a1 = -0.9492274f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (MODE == 0)
  {
    float v0 = u2f(in[0]), v1 = u2f(in[1]);
    for (int k = 0; k < steps; k += 2)
    {
      float r = a1 * y; y = v0 - r;
      r = a1 * y; y = v1 - r;
      asm volatile("" : "+v"(y));
    }
  }
  else
  {
    constexpr int U = 8;
    float ga[U], gb[U];
    auto load_group = [&](float (&g)[U], int at) {
      const uint2 *p2 = reinterpret_cast<const uint2 *>(in + at);
#pragma unroll
      for (int j = 0; j < U / 2; j++) { const uint2 w = p2[j]; g[2 * j] = u2f(w.x); g[2 * j + 1] = u2f(w.y); }
    };
    auto run_group = [&](const float (&g)[U], int at) {
      uint2 *o2 = reinterpret_cast<uint2 *>(o + at);
#pragma unroll
      for (int j = 0; j < U / 2; j++)
      {
        float r = a1 * y; y = g[2 * j] - r; const float y0 = y;
        r = a1 * y; y = g[2 * j + 1] - r;
        if (MODE == 2) o2[j] = make_uint2(f2u(y0), f2u(y));
      }
    };
    load_group(ga, 0);
    int kk = 0;
    for (; kk + 3 * U <= steps; kk += 2 * U)
    {
      load_group(gb, kk + U);
      run_group(ga, kk);
      load_group(ga, kk + 2 * U);
      run_group(gb, kk + U);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) { out[0] = t1 - t0; out[2] = f2u(y); }
}

template <int MODE> static void run(const char *what, int threads, int busy)
{
  unsigned long long *d, h[4];
  hipMalloc(&d, 32);
  const int steps = 256;
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, steps, 274, busy);
  hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
  printf("%-28s %4d threads, other waves %s: %6.2f ticks/step\n", what, threads, busy ? "busy" : "idle", (double)h[0] / steps);
  hipFree(d);
}

int main()
{
  run<0>("chain only", 64, 0);
  run<1>("chain + ds_read2_b64", 64, 0);
  run<2>("chain + reads + writes", 64, 0);
  run<2>("chain + reads + writes", 1024, 0);
  run<0>("chain only", 1024, 1);
  run<2>("chain + reads + writes", 1024, 1);
  return 0;
}
