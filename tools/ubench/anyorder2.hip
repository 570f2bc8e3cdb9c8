// Follow-up to anyorder.hip: does ANY shape of kernel overlap its successor in the same stream under hipExtAnyOrderLaunch?
// Every workgroup records its start and end (s_memrealtime, 100 MHz); per pair of consecutive launches the host prints
// (first start of launch i+1) - (last end of launch i): negative = overlap.  Shapes: the flow kernels' (1024 threads,
// 100 KB of LDS: one workgroup per CU) and a light one (256 threads, no LDS).
// build: hipcc --offload-arch=gfx950 -O3 -o anyorder2 anyorder2.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int LDS_DW>
__global__ void k_busy(unsigned long long *out, int base_us)
{
  __shared__ unsigned int big[LDS_DW];
  big[threadIdx.x % LDS_DW] = blockIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long want = (unsigned long long)(base_us + 2 * (blockIdx.x % 16)) * 100ull;
  while (__builtin_amdgcn_s_memrealtime() - t0 < want)
  {
    __builtin_amdgcn_s_sleep(16);
  }
  if (threadIdx.x == 0)
  {
    out[2 * blockIdx.x] = t0 + (big[7 % LDS_DW] & 0u);
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
}

int main()
{
  const int K = 12, G = 256;
  unsigned long long *d;
  hipMalloc(&d, sizeof(unsigned long long) * 2 * G * K);
  hipStream_t s;
  hipStreamCreate(&s);
  for (int shape = 0; shape < 2; shape++)
  {
    for (int any = 0; any < 2; any++)
    {
      hipMemset(d, 0, sizeof(unsigned long long) * 2 * G * K);
      hipStreamSynchronize(s);
      for (int i = 0; i < K; i++)
      {
        unsigned long long *o = d + (size_t)2 * G * i;
        const unsigned flags = any ? hipExtAnyOrderLaunch : 0u;
        if (shape == 0) hipExtLaunchKernelGGL((k_busy<25000>), dim3(G), dim3(1024), 0, s, nullptr, nullptr, flags, o, 100);
        else hipExtLaunchKernelGGL((k_busy<64>), dim3(G), dim3(256), 0, s, nullptr, nullptr, flags, o, 100);
      }
      hipStreamSynchronize(s);
      std::vector<unsigned long long> h((size_t)2 * G * K);
      hipMemcpy(h.data(), d, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
      printf("%s, %s: gap between launches (first start of i+1 - last end of i), us:", shape == 0 ? "1024 threads + 100 KB LDS" : "256 threads, no LDS",
             any ? "any-order" : "barrier  ");
      for (int i = 0; i + 1 < K; i++)
      {
        unsigned long long last_end = 0, first_start = ~0ull;
        for (int g = 0; g < G; g++)
        {
          last_end = std::max(last_end, h[(size_t)2 * G * i + 2 * g + 1]);
          first_start = std::min(first_start, h[(size_t)2 * G * (i + 1) + 2 * g]);
        }
        printf(" %.1f", ((double)first_start - (double)last_end) / 100.0);
      }
      printf("\n");
    }
  }
  printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
