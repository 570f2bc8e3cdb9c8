// What does a 16-byte-per-lane store cost the SIMD that issues it, by the FORM of its address?  k_mod adds ~0.19 ms of
// stores to 0.66 ms of arithmetic instead of hiding them, even when the stores stay in the L2s: one 1 KiB
// store per ~85 vector instructions of a wave costs its SIMD ~95 cycles.  If that is the store's operands going through
// the vector register file's ports (four data dwords + the address per lane), an address that needs no vector register
// should make it cheaper:
//   form 0  no store (the arithmetic alone)
//   form 1  global_store_dwordx4, 64-bit address per lane
//   form 2  global_store_dwordx4, scalar base + 32-bit offset per lane        (what the compiler makes of k_mod's store)
//   form 3  buffer_store_dwordx4, offen (32-bit offset per lane)
//   form 4  buffer_store_dwordx4 with ADD_TID_ENABLE in the resource: NO vector address (lane l writes base + soffset + 16 l)
//   form 5 / 6  the same bytes as two 8-byte / four 4-byte stores per lane (is the cost per instruction or per byte?)
// The footprint is 16 MiB (stays in the L2s), 131072 workgroups x 8 rounds x 256 lanes like k_mod's 1024 x 16 launch.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o store_issue store_issue.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int TAP>
__device__ __forceinline__ uint32_t mul24(const uint32_t x)
{
  uint32_t r;
  asm("v_mul_i32_i24_e32 %0, %1, %2" : "=v"(r) : "i"(TAP), "v"(x));
  return r;
}

template <int FORM, int WORK>
__global__ __launch_bounds__(256) void k_issue(uint8_t *out, uint32_t window_chunks, uint32_t seed, unsigned long long *clk)
{
  // shader clock (s_memtime) against the constant 100 MHz clock (s_memrealtime), one lane of every 1024th workgroup
  const bool probe = threadIdx.x == 0 && (blockIdx.x & 1023u) == 512u;
  unsigned long long c0 = 0, r0 = 0;
  if (probe)
  {
    c0 = __builtin_readcyclecounter();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  const uint32_t ch = blockIdx.x % window_chunks;           // 32 KiB chunk inside the window
  uint8_t *base = out + (size_t)ch * 32768;
  const uint32_t tid = threadIdx.x, wave = tid >> 6;
  // buffer resource over the whole window (forms 3, 4)
  const uint64_t b = (uint64_t)(uintptr_t)base;
  u32x4 rs;
  rs.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
  rs.y = ((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) & 0xffffu) | (FORM == 4 ? (16u << 16) : 0u);
  rs.z = (FORM == 4) ? 0xffffffffu : 32768u;
  rs.w = (FORM == 4) ? 0x00800000u : 0x00020000u;           // ADD_TID_ENABLE (data format = stride's high bits: 0) | 32-bit raw
  uint32_t a0 = tid * 0x9E3779B1u + seed, a1 = a0 ^ 0x85EBCA77u, a2 = a0 + 0xC2B2AE3Du, a3 = ~a0;
#pragma unroll 1
  for (int r = 0; r < 8; r++)
  {
    // the tail's instruction count, roughly its mix (24-bit multiply-adds, adds, shifts, byte permutes)
#pragma unroll
    for (int k = 0; k < WORK / 8; k++)
    {
      a0 = mul24<19661>(a0) + a1;
      a1 = (a1 + a2) >> 1;
      a2 = __builtin_amdgcn_perm(a2, a3, 0x06020400u) + a0;
      a3 = (a3 ^ a1) + (uint32_t)k;
      a0 += a3;
      a1 = mul24<4551>(a1) + a2;
      a2 = (a2 >> 3) + a1;
      a3 = a3 + a0;
    }
    // (the pattern the host checks, tied to the arithmetic so that none of it is dead)
    const uint32_t never = ((a0 ^ a1) == 0x12345678u && (a2 ^ a3) == 0x9abcdef0u) ? 1u : 0u;
    const uint32_t d0 = ch + never, d1 = (uint32_t)r, d2 = tid, d3 = 7u;
    const uint32_t off = ((uint32_t)r * 256u + tid) * 16u;
    if (FORM == 1)
    {
      uint8_t *p = base + off;
      asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(u32x4{d0, d1, d2, d3}) : "memory");
    }
    else if (FORM == 2)
    {
      asm volatile("global_store_dwordx4 %0, %1, %2 nt" : : "v"(off), "v"(u32x4{d0, d1, d2, d3}), "s"(base) : "memory");
    }
    else if (FORM == 3)
    {
      asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen nt" : : "v"(u32x4{d0, d1, d2, d3}), "v"(off), "s"(rs) : "memory");
    }
    else if (FORM == 4)
    {
      const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((uint32_t)r * 256u + wave * 64u) * 16u));
      asm volatile("buffer_store_dwordx4 %0, off, %1, %2 nt" : : "v"(u32x4{d0, d1, d2, d3}), "s"(rs), "s"(soff) : "memory");
    }
    else if (FORM == 5)
    {
      // the same 16 bytes per lane as TWO 8-byte stores, each wave instruction 512 contiguous bytes (k_rx_wbfm_flow's iq dump)
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      const uint32_t wbase = ((uint32_t)r * 256u + wave * 64u) * 16u, lane = tid & 63u;
      const bool hi = lane >= 32u;                          // (host pattern: lane l of the first store is cell l / 2, half l % 2)
      (void)hi;
      asm volatile("global_store_dwordx2 %0, %1, %2 nt" : : "v"(wbase + lane * 8u), "v"(u32x2{d0, d1}), "s"(base) : "memory");
      asm volatile("global_store_dwordx2 %0, %1, %2 nt" : : "v"(wbase + 512u + lane * 8u), "v"(u32x2{d2, d3}), "s"(base) : "memory");
    }
    else if (FORM == 6)
    {
      // ... as FOUR 4-byte stores, 256 contiguous bytes each
      const uint32_t wbase = ((uint32_t)r * 256u + wave * 64u) * 16u, lane = tid & 63u;
      asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(wbase + lane * 4u), "v"(d0), "s"(base) : "memory");
      asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(wbase + 256u + lane * 4u), "v"(d1), "s"(base) : "memory");
      asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(wbase + 512u + lane * 4u), "v"(d2), "s"(base) : "memory");
      asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(wbase + 768u + lane * 4u), "v"(d3), "s"(base) : "memory");
    }
    else if (never != 0u)
    {
      *reinterpret_cast<u32x4 *>(base + off) = u32x4{d0, d1, d2, d3};   // (never: keeps the arithmetic alive)
    }
  }
  if (probe)
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    atomicAdd(&clk[0], __builtin_readcyclecounter() - c0);
    atomicAdd(&clk[1], __builtin_amdgcn_s_memrealtime() - r0);
  }
}

template <int FORM, int WORK>
static void run(const char *name, uint8_t *win, uint8_t *alloc, size_t alloc_bytes, size_t guard, uint32_t window_chunks)
{
  static unsigned long long *clk = nullptr;
  if (clk == nullptr) hipMalloc(&clk, 16);
  hipMemset(clk, 0, 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipMemset(alloc, 0xA5, alloc_bytes);
  float sum = 0;
  for (int rep = 0; rep < 30; rep++)
  {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_issue<FORM, WORK>), dim3(131072), dim3(256), 0, 0, win, window_chunks, (uint32_t)rep, clk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 10) sum += ms;
  }
  // what landed where?
  std::vector<uint32_t> h(alloc_bytes / 4);
  hipMemcpy(h.data(), alloc, alloc_bytes, hipMemcpyDeviceToHost);
  size_t bad = 0, guard_hit = 0;
  for (size_t i = 0; i < guard / 4; i++) guard_hit += (h[i] != 0xA5A5A5A5u) + (h[h.size() - 1 - i] != 0xA5A5A5A5u);
  if (FORM != 0 && FORM < 5)
  {
    const uint32_t *w = h.data() + guard / 4;
    for (uint32_t ch = 0; ch < window_chunks; ch++)
      for (uint32_t r = 0; r < 8; r++)
        for (uint32_t t = 0; t < 256; t++)
        {
          const uint32_t *q = w + ((size_t)ch * 32768 + ((size_t)r * 256 + t) * 16) / 4;
          bad += !(q[0] == ch && q[1] == r && q[2] == t && q[3] == 7u);
        }
  }
  unsigned long long hc[2];
  hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
  printf("%-72s %.4f ms   shader clock %.0f MHz   wrong cells %zu, guard words touched %zu\n", name, sum / 20, hc[1] ? 100.0 * (double)hc[0] / (double)hc[1] : 0.0, bad,
         guard_hit);
  fflush(stdout);
}

int main()
{
  const uint32_t window_chunks = 512;                       // 16 MiB
  const size_t guard = 1 << 20, win_bytes = (size_t)window_chunks * 32768, alloc_bytes = win_bytes + 2 * guard;
  uint8_t *alloc;
  hipMalloc(&alloc, alloc_bytes);
  uint8_t *win = alloc + guard;
  run<0, 88>("no store, 88 vector instructions per round", win, alloc, alloc_bytes, guard, window_chunks);
  run<1, 88>("global_store_dwordx4, 64-bit address per lane", win, alloc, alloc_bytes, guard, window_chunks);
  run<2, 88>("global_store_dwordx4, scalar base + 32-bit offset per lane", win, alloc, alloc_bytes, guard, window_chunks);
  run<3, 88>("buffer_store_dwordx4 offen", win, alloc, alloc_bytes, guard, window_chunks);
  run<4, 88>("buffer_store_dwordx4, ADD_TID_ENABLE, no vector address", win, alloc, alloc_bytes, guard, window_chunks);
  run<5, 88>("two global_store_dwordx2 (512 contiguous bytes each)", win, alloc, alloc_bytes, guard, window_chunks);
  run<6, 88>("four global_store_dword (256 contiguous bytes each)", win, alloc, alloc_bytes, guard, window_chunks);
  run<0, 8>("no store, 8 vector instructions per round", win, alloc, alloc_bytes, guard, window_chunks);
  run<2, 8>("scalar base + 32-bit offset, 8 vector instructions per round", win, alloc, alloc_bytes, guard, window_chunks);
  run<4, 8>("ADD_TID_ENABLE, 8 vector instructions per round", win, alloc, alloc_bytes, guard, window_chunks);
  run<5, 8>("two dwordx2 stores, 8 vector instructions per round", win, alloc, alloc_bytes, guard, window_chunks);
  run<6, 8>("four dword stores, 8 vector instructions per round", win, alloc, alloc_bytes, guard, window_chunks);
  return 0;
}
