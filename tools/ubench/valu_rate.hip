// Microbenchmark: issue cost of VALU instruction kinds on gfx950 (cycles per
// wave64 instruction, one wave per SIMD and 8 waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REP16(x) x x x x x x x x x x x x x x x x

#define KERNEL(name, body)                                                        \
  __global__ void name(unsigned long long *out, int iters)                        \
  {                                                                               \
    unsigned a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3; \
    unsigned b0 = 0x00010002, b1 = 0x00030004;                                    \
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                     \
    unsigned long long t0 = __builtin_readcyclecounter();                         \
    for (int i = 0; i < iters; i++)                                               \
    {                                                                             \
      REP16(body)                                                                 \
    }                                                                             \
    unsigned long long t1 = __builtin_readcyclecounter();                         \
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                     \
    if ((threadIdx.x & 63) == 0)                                                  \
    {                                                                             \
      out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;           \
      out[8192 + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r1 - r0;    \
    }                                                                             \
    if (a0 + a1 + a2 + a3 == 0x12345678) out[0] = b0 + b1;                        \
  }

// 4 independent chains per body -> 64 instructions per loop iteration
KERNEL(k_add_u32, asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
KERNEL(k_pk_add_u16, asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
KERNEL(k_pk_mad_u16, asm volatile("v_pk_mad_u16 %0, %0, %4, %5\n v_pk_mad_u16 %1, %1, %4, %5\n v_pk_mad_u16 %2, %2, %4, %5\n v_pk_mad_u16 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
KERNEL(k_pk_ashr, asm volatile("v_pk_ashrrev_i16 %0, 1, %0\n v_pk_ashrrev_i16 %1, 1, %1\n v_pk_ashrrev_i16 %2, 1, %2\n v_pk_ashrrev_i16 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_perm, asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
KERNEL(k_dpp_shr, asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_dpp_row, asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_mad_u24, asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
KERNEL(k_dot2, asm volatile("v_dot2_i32_i16 %0, %4, %5, %0\n v_dot2_i32_i16 %1, %4, %5, %1\n v_dot2_i32_i16 %2, %4, %5, %2\n v_dot2_i32_i16 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
KERNEL(k_mul_f32, asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
KERNEL(k_lshl_add, asm volatile("v_lshl_add_u32 %0, %0, 1, %4\n v_lshl_add_u32 %1, %1, 1, %4\n v_lshl_add_u32 %2, %2, 1, %4\n v_lshl_add_u32 %3, %3, 1, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
KERNEL(k_add3, asm volatile("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
KERNEL(k_sad_u8, asm volatile("v_sad_u8 %0, %0, %4, %5\n v_sad_u8 %1, %1, %4, %5\n v_sad_u8 %2, %2, %4, %5\n v_sad_u8 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
KERNEL(k_add_sdwa, asm volatile("v_add_u32_sdwa %0, %0, %4 dst_sel:DWORD src0_sel:BYTE_0 src1_sel:BYTE_2\n v_add_u32_sdwa %1, %1, %4 dst_sel:DWORD src0_sel:BYTE_0 src1_sel:BYTE_2\n v_add_u32_sdwa %2, %2, %4 dst_sel:DWORD src0_sel:BYTE_0 src1_sel:BYTE_2\n v_add_u32_sdwa %3, %3, %4 dst_sel:DWORD src0_sel:BYTE_0 src1_sel:BYTE_2" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
// dependent chain of v_add_u32 (latency)
KERNEL(k_dep_add, asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %0, %0, %4\n v_add_u32 %0, %0, %4\n v_add_u32 %0, %0, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
KERNEL(k_dep_mulf, asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %0, %0, %4\n v_mul_f32 %0, %0, %4\n v_mul_f32 %0, %0, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
KERNEL(k_dep_pk, asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %0, %0, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)

// literal operands (64-bit encodings of VOP2), scalar operands, and the other instruction kinds of the WBFM chunk loop
KERNEL(k_add_lit, asm volatile("v_add_u32 %0, 0x12345678, %0\n v_add_u32 %1, 0x12345678, %1\n v_add_u32 %2, 0x12345678, %2\n v_add_u32 %3, 0x12345678, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_and_lit, asm volatile("v_and_b32 %0, 0x00ff00ff, %0\n v_and_b32 %1, 0x00ff00ff, %1\n v_and_b32 %2, 0x00ff00ff, %2\n v_and_b32 %3, 0x00ff00ff, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_add_sgpr, asm volatile("v_add_u32 %0, %4, %0\n v_add_u32 %1, %4, %1\n v_add_u32 %2, %4, %2\n v_add_u32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(iters));)
KERNEL(k_add_inline, asm volatile("v_add_u32 %0, 17, %0\n v_add_u32 %1, 17, %1\n v_add_u32 %2, 17, %2\n v_add_u32 %3, 17, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_fmaak, asm volatile("v_fmaak_f32 %0, %0, %4, 0x3d27d12b\n v_fmaak_f32 %1, %1, %4, 0x3d27d12b\n v_fmaak_f32 %2, %2, %4, 0x3d27d12b\n v_fmaak_f32 %3, %3, %4, 0x3d27d12b" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
KERNEL(k_fmac, asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
KERNEL(k_fma_e64, asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
KERNEL(k_cndmask_vcc, asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc");)
KERNEL(k_cndmask_e64, asm volatile("v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cndmask_b32 %1, %1, %4, s[20:21]\n v_cndmask_b32 %2, %2, %4, s[20:21]\n v_cndmask_b32 %3, %3, %4, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "s20", "s21");)
KERNEL(k_bfe, asm volatile("v_bfe_i32 %0, %0, %4, 2\n v_bfe_i32 %1, %1, %4, 2\n v_bfe_i32 %2, %2, %4, 2\n v_bfe_i32 %3, %3, %4, 2" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
KERNEL(k_lerp, asm volatile("v_lerp_u8 %0, %0, %4, %5\n v_lerp_u8 %1, %1, %4, %5\n v_lerp_u8 %2, %2, %4, %5\n v_lerp_u8 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
KERNEL(k_cvt_f32_u32, asm volatile("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_lshr, asm volatile("v_lshrrev_b32 %0, 7, %0\n v_lshrrev_b32 %1, 7, %1\n v_lshrrev_b32 %2, 7, %2\n v_lshrrev_b32 %3, 7, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_mul_u24, asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
KERNEL(k_cmp_e32, asm volatile("v_cmp_gt_u32 vcc, %0, %4\n v_cmp_gt_u32 vcc, %1, %4\n v_cmp_gt_u32 vcc, %2, %4\n v_cmp_gt_u32 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc");)
KERNEL(k_xad, asm volatile("v_xad_u32 %0, %0, %4, %5\n v_xad_u32 %1, %1, %4, %5\n v_xad_u32 %2, %2, %4, %5\n v_xad_u32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)

// mixed streams: do the fast-class instructions keep their rate between slow-class ones?
KERNEL(k_mix_add_perm, asm volatile("v_add_u32 %0, %0, %4\n v_perm_b32 %1, %1, %4, %5\n v_add_u32 %2, %2, %4\n v_perm_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
KERNEL(k_mix_3add_perm, asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_perm_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
KERNEL(k_mix_fma_dpp, asm volatile("v_fmac_f32 %0, %4, %5\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32 %2, %4, %5\n v_mov_b32_dpp %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
KERNEL(k_mix_dep, asm volatile("v_add_u32 %0, %0, %4\n v_perm_b32 %0, %0, %4, %5\n v_add_u32 %0, %0, %4\n v_perm_b32 %0, %0, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)

// v_cndmask with the mask in VCC: alone, behind the compare that writes VCC, and in the 64-bit encoding
KERNEL(k_cmp_cnd_vcc, asm volatile("v_cmp_gt_u32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_gt_u32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc");)
KERNEL(k_cnd_vcc_e64, asm volatile("v_cndmask_b32_e64 %0, %0, %4, vcc\n v_cndmask_b32_e64 %1, %1, %4, vcc\n v_cndmask_b32_e64 %2, %2, %4, vcc\n v_cndmask_b32_e64 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc");)
KERNEL(k_cmp_cnd_sgpr, asm volatile("v_cmp_gt_u32 s[20:21], %0, %4\n v_cndmask_b32 %1, %1, %4, s[20:21]\n v_cmp_gt_u32 s[22:23], %2, %4\n v_cndmask_b32 %3, %3, %4, s[22:23]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "s20", "s21", "s22", "s23");)

typedef void (*kern_t)(unsigned long long *, int);
// grid = 512 workgroups of 1024 threads = 8 waves on every SIMD of the chip; wall clock by
// HIP events -> ns per wave-instruction per SIMD (a SIMD16 at 2.4 GHz would give 1.67 ns)
static double run(kern_t k, int threads, int grid)
{
  static unsigned long long h[16384];
  unsigned long long *d;
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
  const double ghz = sum / rsum * 0.1;                       // s_memtime ticks per 100 MHz s_memrealtime tick
  hipFree(d);
  return ms * 1e6 / instr_per_simd * ghz;                    // shader cycles per wave64 instruction per SIMD
}
static void run4(const char *name, kern_t k)
{
  printf("%-14s cycles per wave64 instruction per SIMD at 1 / 2 / 4 / 8 waves per SIMD: %5.2f %5.2f %5.2f %5.2f\n", name,
         run(k, 256, 256), run(k, 512, 256), run(k, 1024, 256), run(k, 1024, 512));
}
#define RUN(k) run4(#k, k);
int main()
{
  RUN(k_add_u32) RUN(k_pk_add_u16) RUN(k_pk_mad_u16) RUN(k_pk_ashr) RUN(k_perm) RUN(k_dpp_shr) RUN(k_dpp_row)
  RUN(k_mad_u24) RUN(k_dot2) RUN(k_mul_f32) RUN(k_lshl_add) RUN(k_add3) RUN(k_sad_u8) RUN(k_add_sdwa)
  RUN(k_dep_add) RUN(k_dep_mulf) RUN(k_dep_pk)
  RUN(k_add_lit) RUN(k_and_lit) RUN(k_add_sgpr) RUN(k_add_inline) RUN(k_fmaak) RUN(k_fmac) RUN(k_fma_e64) RUN(k_cndmask_vcc) RUN(k_cndmask_e64)
  RUN(k_mix_add_perm) RUN(k_mix_3add_perm) RUN(k_mix_fma_dpp) RUN(k_mix_dep)
  RUN(k_cmp_cnd_vcc) RUN(k_cnd_vcc_e64) RUN(k_cmp_cnd_sgpr)
  RUN(k_bfe) RUN(k_lerp) RUN(k_cvt_f32_u32) RUN(k_lshr) RUN(k_mul_u24) RUN(k_cmp_e32) RUN(k_xad)
  return 0;
}
