// Do consecutive kernels of ONE stream overlap when they are launched with hipExtAnyOrderLaunch (no barrier bit in the
// AQL packet)?  256 workgroups x 1024 threads, workgroup i busy for 100 + 2 (i % 16) us (s_memrealtime, 100 MHz);
// K launches back to back: with the barrier the time is K x (130 us + launch gap), without it K x ~115 us if the
// dispatcher lets the next kernel's workgroups onto CUs as they free up.
// build: hipcc --offload-arch=gfx950 -O3 -o anyorder anyorder.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>

__global__ __launch_bounds__(1024) void k_busy(unsigned long long *out, int base_us)
{
  __shared__ unsigned int big[25000];                    // 100 KB: one workgroup per CU, like the flow kernels
  big[threadIdx.x] = blockIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long want = (unsigned long long)(base_us + 2 * (blockIdx.x % 16)) * 100ull;
  while (__builtin_amdgcn_s_memrealtime() - t0 < want)
  {
    __builtin_amdgcn_s_sleep(16);
  }
  if (threadIdx.x == 0)
  {
    out[blockIdx.x] = t0 + big[(threadIdx.x * 7) % 25000];
  }
}

int main()
{
  unsigned long long *d;
  hipMalloc(&d, 8 * 256 * 64);
  hipStream_t s;
  hipStreamCreate(&s);
  const int K = 50;
  for (int mode = 0; mode < 4; mode++)
  {
    const bool any = mode & 1;
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k_busy, dim3(256), dim3(1024), 0, s, d, 100);
    hipStreamSynchronize(s);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < K; i++)
    {
      if (any) hipExtLaunchKernelGGL(k_busy, dim3(256), dim3(1024), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d + 256 * (i % 64), 100);
      else hipLaunchKernelGGL(k_busy, dim3(256), dim3(1024), 0, s, d + 256 * (i % 64), 100);
    }
    hipStreamSynchronize(s);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("%s: %d launches, %.1f us per launch (workgroups busy 100..130 us, mean 115)\n", any ? "any-order" : "barrier  ", K, us / K);
  }
  hipError_t e = hipGetLastError();
  printf("last error: %s\n", hipGetErrorString(e));
  return 0;
}
