// Microbenchmark: the real phase-A instruction mix (quad_piece of hrfd_rx_kernels.hip) on register-resident
// inputs, no HBM, at 1 / 2 / 3 / 4 waves per SIMD: cycles per wave64 VALU instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../hackrfdiags_amd/csrc -o quad_rate quad_rate.hip
#include "hrfd_rx_kernels.hip"
#include <stdio.h>
using namespace hrfd;

template <int ATAN>
__global__ __launch_bounds__(1024, 4) void k_quad(unsigned long long *out, int iters, const uint8_t *gcorr, const float *ginv, float kgain)
{
  __shared__ __attribute__((aligned(16))) uint8_t atcorr[kCorrBytes];
  __shared__ __attribute__((aligned(16))) float atinv[ATAN == 2 ? kCorrBytes : kInvEntries];
  __shared__ uint32_t sink[1024 * 4];
  for (int i = threadIdx.x; i < kCorrBytes; i += blockDim.x) atcorr[i] = gcorr[i];
  for (int i = threadIdx.x; i < (ATAN == 2 ? kCorrBytes : kInvEntries); i += blockDim.x) atinv[i] = ginv[i % kInvEntries];
  __syncthreads();
  StreamCtx X;
  X.kgain = kgain;
  X.atc = atcorr;
  X.ati = atinv;
  X.lane = threadIdx.x & 63;
  QuadCarry c;
  c.fe = {0x80808080u, 0x00800080u, 0x00800080u};
  c.theta = 0u;
  c.p = 0u;
  uint4 raw[4];
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
  for (int j = 0; j < 4; j++)
  {
    s = s * 1664525u + 1013904223u; raw[j].x = s;
    s = s * 1664525u + 1013904223u; raw[j].y = s;
    s = s * 1664525u + 1013904223u; raw[j].z = s;
    s = s * 1664525u + 1013904223u; raw[j].w = s;
  }
  uint32_t acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++)
  {
    uint32_t v[4], mag4;
    float theta[4];
    quad_piece<ATAN>(raw, c, X, v, theta, mag4);
    // feed the outputs back so that nothing is hoisted or dropped; one LDS store like the kernel's
    *reinterpret_cast<uint4 *>(sink + 4 * threadIdx.x) = make_uint4(v[0], v[1], v[2], v[3]);
    acc += mag4;
    raw[0].x += v[0] | 1u; raw[1].y ^= v[1]; raw[2].z += v[2]; raw[3].w ^= v[3] + acc;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0)
  {
    out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  }
  if (acc == 0x12345678u) out[0] = sink[threadIdx.x];
}

int main()
{
  static unsigned long long h[8192];
  unsigned long long *d;
  uint8_t *dc; float *di;
  hipMalloc(&d, sizeof(h));
  hipMalloc(&dc, kCorrBytes); hipMalloc(&di, sizeof(float) * kInvEntries);
  hipMemset(dc, 0x55, kCorrBytes);
  float inv[kInvEntries] = {0};
  for (int a = 1; a <= 128; a++) inv[a] = 1.0f / (float)a;
  hipMemcpy(di, inv, sizeof(inv), hipMemcpyHostToDevice);
  const int iters = 2000;
  for (int atan = 1; atan <= 2; atan++)
  for (int threads : {256, 512, 768, 1024})
  {
    auto kern = (atan == 1) ? k_quad<1> : k_quad<2>;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, iters, dc, di, 13351.9f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, iters, dc, di, 13351.9f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0; int n = 256 * (threads / 64);
    for (int i = 0; i < n; i++) sum += (double)h[i];
    const double cyc = sum / n;                      // cycles one wave spent on `iters` pieces
    printf("atan %d, %d waves/SIMD: %.0f cycles per piece per wave, %.0f cycles per piece per SIMD (wall %.3f ms)\n",
           atan, threads / 256, cyc / iters, cyc / iters / (threads / 256), ms);
  }
  return 0;
}
