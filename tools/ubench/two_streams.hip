// Do kernels of two HIP streams overlap on this box?  A one-workgroup kernel that spins for ~2 ms on each of N
// streams, forked from and joined to a third stream with events (the pattern of a bank cut into groups).
// Build: hipcc --offload-arch=gfx950 -O3 -o two_streams two_streams.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_spin(unsigned long long ticks, unsigned *sink)
{
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned n = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) n++;
  if (threadIdx.x == 0) sink[blockIdx.x] = n;
}
__global__ void k_bulk(float *p, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.0f;
}
int main()
{
  unsigned *sink; float *buf;
  const size_t n = 256u << 20;
  hipMalloc(&sink, 4096); hipMalloc(&buf, n * 4);
  hipStream_t s, q[4]; hipEvent_t fork, join[4], t0, t1;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int i = 0; i < 4; i++) { hipStreamCreateWithFlags(&q[i], hipStreamNonBlocking); hipEventCreateWithFlags(&join[i], hipEventDisableTiming); }
  hipEventCreateWithFlags(&fork, hipEventDisableTiming); hipEventCreate(&t0); hipEventCreate(&t1);
  for (int groups = 1; groups <= 4; groups++)
  {
    for (int rep = 0; rep < 3; rep++)
    {
      hipEventRecord(t0, s);
      hipEventRecord(fork, s);
      for (int g = 0; g < groups; g++)
      {
        hipStreamWaitEvent(q[g], fork, 0);
        hipLaunchKernelGGL(k_bulk, dim3(2048), dim3(256), 0, q[g], buf + (size_t)g * (n / 4), n / 4);
        hipLaunchKernelGGL(k_spin, dim3(4), dim3(448), 0, q[g], 200000ull, sink + 16 * g);      // 2 ms
        hipLaunchKernelGGL(k_bulk, dim3(2048), dim3(256), 0, q[g], buf + (size_t)g * (n / 4), n / 4);
        hipEventRecord(join[g], q[g]);
        hipStreamWaitEvent(s, join[g], 0);
      }
      hipEventRecord(t1, s);
      hipStreamSynchronize(s);
      float ms; hipEventElapsedTime(&ms, t0, t1);
      if (rep == 2) printf("%d group(s), each bulk + 2 ms spin + bulk on its own stream: %.3f ms\n", groups, ms);
    }
  }
  return 0;
}
