// Microbenchmark (round 4): issue cost and semantics of v_mqsad_pk_u16_u8 (four masked byte SADs in one instruction:
// with the mask word 0x00000080 it is |byte - 128| of four bytes at once, i.e. |i|, |q| of two offset-binary samples),
// and of the packed f32 instructions, against v_perm_b32.  cycles per wave64 instruction per SIMD at 1/2/4/8 waves.
// Build: hipcc --offload-arch=gfx950 -O3 -o mqsad_rate mqsad_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP16(x) x x x x x x x x x x x x x x x x
typedef unsigned long long u64;

#define KERNEL(name, decl, body, sink)                                            \
  __global__ void name(u64 *out, int iters)                                       \
  {                                                                               \
    decl                                                                          \
    u64 r0 = __builtin_amdgcn_s_memrealtime();                                    \
    u64 t0 = __builtin_readcyclecounter();                                        \
    for (int i = 0; i < iters; i++)                                               \
    {                                                                             \
      REP16(body)                                                                 \
    }                                                                             \
    u64 t1 = __builtin_readcyclecounter();                                        \
    u64 r1 = __builtin_amdgcn_s_memrealtime();                                    \
    if ((threadIdx.x & 63) == 0)                                                  \
    {                                                                             \
      out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;           \
      out[8192 + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r1 - r0;    \
    }                                                                             \
    if (sink) out[0] = 1;                                                         \
  }

KERNEL(k_perm,
       unsigned a0 = threadIdx.x; unsigned a1 = a0 * 3 + 1; unsigned a2 = a0 * 5 + 2; unsigned a3 = a0 * 7 + 3; unsigned b0 = 0x00010002; unsigned b1 = 0x00030004;,
       asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));,
       a0 + a1 + a2 + a3 == 0x12345678)
KERNEL(k_mqsad,
       u64 s0 = threadIdx.x * 0x0101010101010101ull; u64 s1 = s0 + 7; u64 d0 = 0; u64 d1 = 0; u64 d2 = 0; u64 d3 = 0; unsigned m = 0x00000080;,
       asm volatile("v_mqsad_pk_u16_u8 %0, %4, %6, %0\n v_mqsad_pk_u16_u8 %1, %5, %6, %1\n v_mqsad_pk_u16_u8 %2, %4, %6, %2\n v_mqsad_pk_u16_u8 %3, %5, %6, %3" : "+&v"(d0), "+&v"(d1), "+&v"(d2), "+&v"(d3) : "v"(s0), "v"(s1), "v"(m));,
       d0 + d1 + d2 + d3 == 0x12345678)
KERNEL(k_pk_fma_f32,
       float2 a0 = make_float2(threadIdx.x, 1.0f); float2 a1 = a0; float2 a2 = a0; float2 a3 = a0; float2 b = make_float2(1.0001f, 0.9999f); float2 c = make_float2(0.5f, 0.25f);,
       asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));,
       a0.x + a1.x + a2.y + a3.y == 123.0f)
KERNEL(k_pk_add_f32,
       float2 a0 = make_float2(threadIdx.x, 1.0f); float2 a1 = a0; float2 a2 = a0; float2 a3 = a0; float2 b = make_float2(1.0001f, 0.9999f);,
       asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));,
       a0.x + a1.x + a2.y + a3.y == 123.0f)
KERNEL(k_pk_lshr_b16,
       unsigned a0 = threadIdx.x; unsigned a1 = a0 * 3 + 1; unsigned a2 = a0 * 5 + 2; unsigned a3 = a0 * 7 + 3;,
       asm volatile("v_pk_lshrrev_b16 %0, 1, %0\n v_pk_lshrrev_b16 %1, 1, %1\n v_pk_lshrrev_b16 %2, 1, %2\n v_pk_lshrrev_b16 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));,
       a0 + a1 + a2 + a3 == 0x12345678)

__global__ void k_sem(u64 *out, const u64 *in)
{
  u64 s0 = in[threadIdx.x], acc = 0, d;
  unsigned m = 0x00000080;
  asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %3" : "=&v"(d) : "v"(s0), "v"(m), "v"(acc));
  out[threadIdx.x] = d;
}

typedef void (*kern_t)(u64 *, int);
static double run(kern_t k, int threads, int grid)
{
  static u64 h[16384];
  u64 *d;
  hipMalloc(&d, sizeof(h));
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double sum = 0, rsum = 0; int n = grid * (threads / 64);
  for (int i = 0; i < n; i++) { sum += (double)h[i]; rsum += (double)h[8192 + i]; }
  const double waves_per_simd = (double)grid * (threads / 64) / 1024.0;
  const double instr_per_simd = waves_per_simd * iters * 64.0;
  const double ghz = sum / rsum * 0.1;
  hipFree(d);
  return ms * 1e6 / instr_per_simd * ghz;
}
#define RUN(k) printf("%-14s cycles per wave64 instruction per SIMD at 1 / 2 / 4 / 8 waves per SIMD: %5.2f %5.2f %5.2f %5.2f\n", #k, run(k, 256, 256), run(k, 512, 256), run(k, 1024, 256), run(k, 1024, 512));
int main()
{
  u64 hin[64], hout[64], *din, *dout;
  for (int i = 0; i < 64; i++) hin[i] = 0x00ff7f8081020100ull + (u64)i * 0x0000000100000001ull;
  hipMalloc(&din, sizeof(hin)); hipMalloc(&dout, sizeof(hout));
  hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, dout, din);
  hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
  for (int i = 0; i < 3; i++) printf("mqsad(mask 0x80): S0 = %016llx -> D = %016llx  (expect |byte p - 128| in u16 field p, p = 0..3)\n", hin[i], hout[i]);
  RUN(k_perm) RUN(k_mqsad) RUN(k_pk_fma_f32) RUN(k_pk_add_f32) RUN(k_pk_lshr_b16)
  return 0;
}
