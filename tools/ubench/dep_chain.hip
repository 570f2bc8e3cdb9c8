// Microbenchmark: cycles per instruction of DEPENDENT chains issued by one lone wave (the situation of
// k_phase_scan's recurrence wave), per instruction kind and for the scan's own step.  s_memtime counts at the
// shader clock on this part; the 100 MHz clock (s_memrealtime) is printed beside it.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o dep_chain dep_chain.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP 512
template <int KIND>
__global__ void k_chain(unsigned long long *out, float *sink, float x0, float s0, int iters)
{
  float a = x0 + threadIdx.x * 1e-3f, s = s0;
  const float kM = 0x1.45f308p-3f, kChi = 0x1.921fb6p+2f, kClo = -0x1.777a5cp-23f;
  float rM = kM, rChi = -kChi, rClo = -kClo;             // the constants in registers
  asm volatile("" : "+v"(rM), "+v"(rChi), "+v"(rClo));
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++)
  {
#pragma unroll
    for (int i = 0; i < REP; i++)
    {
      if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(s));
      if (KIND == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(rM));
      if (KIND == 2) asm volatile("v_rndne_f32 %0, %0" : "+v"(a));
      if (KIND == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(s), "v"(rChi));
      if (KIND == 4) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(s), "v"(rChi));
      if (KIND == 5) asm volatile("v_mul_f32 %0, 0x3e22f984, %0" : "+v"(a));     // literal operand
      if (KIND == 6)
      {
        // the scan's step, literal constants (what the compiler emits)
        float k;
        asm volatile("v_add_f32 %0, %0, %2\n\tv_mul_f32 %1, 0x3e22f984, %0\n\tv_rndne_f32 %1, %1\n\t"
                     "v_fmamk_f32 %0, %1, 0xc0c90fdb, %0\n\tv_fmamk_f32 %0, %1, 0x343bbd2e, %0"
                     : "+v"(a), "=&v"(k) : "v"(s));
      }
      if (KIND == 7)
      {
        // the same with the constants in registers
        float k;
        asm volatile("v_add_f32 %0, %0, %2\n\tv_mul_f32 %1, %3, %0\n\tv_rndne_f32 %1, %1\n\t"
                     "v_fmac_f32 %0, %1, %4\n\tv_fmac_f32 %0, %1, %5"
                     : "+v"(a), "=&v"(k) : "v"(s), "v"(rM), "v"(rChi), "v"(rClo));
      }
      if (KIND == 8)
      {
        // magic-number rounding instead of v_rndne: k = (a*M + 1.5*2^23) - 1.5*2^23
        float k;
        asm volatile("v_add_f32 %0, %0, %2\n\tv_fma_f32 %1, %3, %0, %6\n\tv_sub_f32 %1, %1, %6\n\t"
                     "v_fmac_f32 %0, %1, %4\n\tv_fmac_f32 %0, %1, %5"
                     : "+v"(a), "=&v"(k) : "v"(s), "v"(rM), "v"(rChi), "v"(rClo), "v"(12582912.0f));
      }
      if (KIND == 9)
      {
        // two independent chains interleaved (does a second chain fit into the stalls of the first?)
        float k, k2;
        static float b;
        asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %6, %6, %2\n\tv_mul_f32 %1, %3, %0\n\tv_mul_f32 %7, %3, %6\n\t"
                     "v_rndne_f32 %1, %1\n\tv_rndne_f32 %7, %7\n\t"
                     "v_fmac_f32 %0, %1, %4\n\tv_fmac_f32 %6, %7, %4\n\tv_fmac_f32 %0, %1, %5\n\tv_fmac_f32 %6, %7, %5"
                     : "+v"(a), "=&v"(k) : "v"(s), "v"(rM), "v"(rChi), "v"(rClo), "v"(s0), "v"(k2));
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  sink[threadIdx.x] = a;
  if (threadIdx.x == 0)
  {
    out[0] = t1 - t0;
    out[1] = r1 - r0;
  }
}

// the compiled form of the scan's step (what k_phase_scan's recurrence wave runs): 64 steps per loop turn out of
// registers, the phases kept (stored once at the end so that nothing is dropped)
template <int OTHERS>
__global__ void k_compiled(unsigned long long *out, float *sink, const float *steps, float x0, int iters)
{
  __shared__ uint32_t flag;
  if (threadIdx.x == 0) flag = 0;
  __syncthreads();
  if (threadIdx.x >= 64)
  {
    // the other waves of the workgroup poll a flag, like k_phase_scan's movers while they have nothing to do
    while (__hip_atomic_load(&flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) __builtin_amdgcn_s_sleep(1);
    return;
  }
  const float kM = 0x1.45f308p-3f, kChi = 0x1.921fb6p+2f, kClo = -0x1.777a5cp-23f;
  float st[64], ph[64];
  for (int j = 0; j < 64; j++) st[j] = steps[j * 64 + threadIdx.x];
  float a = x0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++)
  {
#pragma unroll
    for (int j = 0; j < 64; j++)
    {
      ph[j] = a;
      a = a + st[j];
      const float k = __builtin_rintf(a * kM);
      a = __builtin_fmaf(k, -kChi, a);
      a = __builtin_fmaf(k, -kClo, a);
    }
    asm volatile("" : "+v"(a));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float acc = a;
  for (int j = 0; j < 64; j++) acc += ph[j];
  sink[threadIdx.x] = acc;
  if (threadIdx.x == 0)
  {
    out[0] = t1 - t0;
    out[1] = r1 - r0;
    __hip_atomic_store(&flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

int main()
{
  unsigned long long *out;
  float *sink;
  hipMalloc(&out, 16);
  hipMalloc(&sink, 64 * 4);
  const char *names[] = {"v_add_f32 (reg)", "v_mul_f32 (reg)", "v_rndne_f32", "v_fma_f32 (VOP3)", "v_fmac_f32 (VOP2)", "v_mul_f32 (literal)",
                         "scan step, literals (5 instr)", "scan step, registers (5 instr)", "scan step, magic rounding (5 instr)"};
  for (int kind = 0; kind < 9; kind++)
  {
    const int iters = 200;
    for (int rep = 0; rep < 2; rep++)
    {
      switch (kind)
      {
        case 0: hipLaunchKernelGGL(k_chain<0>, dim3(1), dim3(64), 0, 0, out, sink, 0.1f, 1e-3f, iters); break;
        case 1: hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(64), 0, 0, out, sink, 0.1f, 1e-3f, iters); break;
        case 2: hipLaunchKernelGGL(k_chain<2>, dim3(1), dim3(64), 0, 0, out, sink, 0.1f, 1e-3f, iters); break;
        case 3: hipLaunchKernelGGL(k_chain<3>, dim3(1), dim3(64), 0, 0, out, sink, 0.1f, 1e-3f, iters); break;
        case 4: hipLaunchKernelGGL(k_chain<4>, dim3(1), dim3(64), 0, 0, out, sink, 0.1f, 1e-3f, iters); break;
        case 5: hipLaunchKernelGGL(k_chain<5>, dim3(1), dim3(64), 0, 0, out, sink, 0.1f, 1e-3f, iters); break;
        case 6: hipLaunchKernelGGL(k_chain<6>, dim3(1), dim3(64), 0, 0, out, sink, 0.1f, 0.7f, iters); break;
        case 7: hipLaunchKernelGGL(k_chain<7>, dim3(1), dim3(64), 0, 0, out, sink, 0.1f, 0.7f, iters); break;
        case 8: hipLaunchKernelGGL(k_chain<8>, dim3(1), dim3(64), 0, 0, out, sink, 0.1f, 0.7f, iters); break;
      }
      hipDeviceSynchronize();
    }
    unsigned long long h[2];
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    const double n = (double)iters * REP;
    printf("%-40s %7.2f cycles (s_memtime) %7.2f ns per chain element\n", names[kind], h[0] / n, h[1] * 10.0 / n);
  }
  float *steps;
  hipMalloc(&steps, 64 * 64 * 4);
  {
    float h[64 * 64];
    for (int i = 0; i < 64 * 64; i++) h[i] = ((i * 7919) % 2000 - 1000) * 1.8e-3f;
    hipMemcpy(steps, h, sizeof(h), hipMemcpyHostToDevice);
  }
  for (int others = 0; others < 2; others++)
  {
    for (int rep = 0; rep < 2; rep++)
    {
      if (others) hipLaunchKernelGGL(k_compiled<1>, dim3(1), dim3(448), 0, 0, out, sink, steps, 0.1f, 2000);
      else hipLaunchKernelGGL(k_compiled<0>, dim3(1), dim3(64), 0, 0, out, sink, steps, 0.1f, 2000);
      hipDeviceSynchronize();
    }
    unsigned long long h[2];
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    const double n = 2000.0 * 64;
    printf("%-40s %7.2f cycles (s_memtime) %7.2f ns per step\n", others ? "compiled step, 6 polling waves beside it" : "compiled step, lone wave", h[0] / n, h[1] * 10.0 / n);
  }
  return 0;
}
