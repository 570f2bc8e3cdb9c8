// The Nco phase recurrence kernels alone on synthetic steps: ms per launch, ns per step, and the result against the reference's loops on
// the host.  (Round 3 also ran a variant here that moved the cells between memory and the recurrence wave's registers
// directly, 16 buffer_load_dwordx4 + 16 buffer_store_dwordx4 per 64 steps and no LDS: 16.91 ns per step against 16.79 --
// what the wave pays beside its chain is the 32 128-bit transfers per chunk themselves, whichever pipeline they use.)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../hackrfdiags_amd/csrc -o phase_scan_rate phase_scan_rate.hip
// Run:   ./phase_scan_rate [channels] [steps] [row_stride] [kernel: 0 = k_phase_rows | 64 = k_phase_scan<64>]
#include "hrfd_rx_kernels.hip"
#include "../../include/hrfd.h"
#include "hrfd_tx_kernels.hip"
#include <stdio.h>
#include <string.h>
#include <vector>
using namespace hrfd;

static float host_wrap(float acc)
{
  const double pi = 3.14159265358979323846, two_pi = 6.283185307179586476925286766559;
  for (int t = 0; t < 64 && (double)acc > pi; t++) acc = (float)((double)acc - two_pi);
  for (int t = 0; t < 64 && (double)acc < -pi; t++) acc = (float)((double)acc + two_pi);
  return acc;
}

// K = 0: k_phase_rows (round 4: lane = time, four channels per wave); K = 64: k_phase_scan<64> (round 2: lane = channel)
static void launch(int K, uint32_t *cells, size_t steps, size_t stride, float *acc, uint32_t C, uint32_t *err)
{
  if (K == 0)
  {
    hipLaunchKernelGGL(k_phase_rows, dim3((C + 15) / 16), dim3(kPrThreads), 0, 0, cells, steps, stride, acc, C);
  }
  else
  {
    hipLaunchKernelGGL((k_phase_scan<64>), dim3((C + 63) / 64), dim3(kPsThreads), 0, 0, cells, steps, stride, acc, C, err);
  }
}

int main(int argc, char **argv)
{
  const uint32_t C = argc > 1 ? atoi(argv[1]) : 1024;
  const size_t steps = argc > 2 ? atol(argv[2]) : 262144;
  const size_t stride = argc > 3 ? atol(argv[3]) : steps;
  const int K = argc > 4 ? atoi(argv[4]) : 0;
  uint32_t *cells, *err;
  float *acc;
  hipMalloc(&cells, (size_t)C * stride * 4);
  hipMalloc(&acc, C * 4);
  hipMalloc(&err, 64);
  std::vector<float> h((size_t)C * stride), want((size_t)C * stride), got((size_t)C * stride);
  unsigned s = 12345;
  for (size_t i = 0; i < h.size(); i++)
  {
    s = s * 1664525u + 1013904223u;
    h[i] = ((int)(s >> 8) % 2000 - 1000) * 1.8e-3f;
  }
  if (C > 3 && steps > 300)
  {
    h[3 * stride + 200] = 31.0f;                         // one absurd step: its chunk takes the loops
  }
  const bool check = (size_t)C * steps <= (size_t)1 << 26;
  std::vector<float> acc_want(C, 0.0f);
  if (check)
  {
    want = h;
    for (uint32_t c = 0; c < C; c++)
    {
      float a = 0.0f;
      for (size_t k = 0; k < steps; k++)
      {
        const float st = h[c * stride + k];
        want[c * stride + k] = a;
        a = host_wrap(a + st);
      }
      acc_want[c] = a;
    }
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int which = 0; which < 1; which++)
  {
    for (int rep = 0; rep < 3; rep++)
    {
      hipMemcpy(cells, h.data(), h.size() * 4, hipMemcpyHostToDevice);
      hipMemset(acc, 0, C * 4);
      hipMemset(err, 0, 64);
      hipEventRecord(e0, 0);
      launch(K, cells, steps, stride, acc, C, err);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      uint32_t he[4];
      hipMemcpy(he, err, 16, hipMemcpyDeviceToHost);
      size_t bad = 0, first = 0;
      if (check && rep == 0)
      {
        hipMemcpy(got.data(), cells, got.size() * 4, hipMemcpyDeviceToHost);
        std::vector<float> ga(C);
        hipMemcpy(ga.data(), acc, C * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < got.size(); i++)
        {
          if (memcmp(&got[i], &want[i], 4) != 0)
          {
            if (!bad) first = i;
            bad++;
          }
        }
        for (uint32_t c = 0; c < C; c++)
        {
          if (memcmp(&ga[c], &acc_want[c], 4) != 0)
          {
            if (!bad) first = (size_t)c * stride + steps;
            bad++;
          }
        }
      }
      printf("%s (%d) %u channels x %zu steps (rows %zu apart): %.3f ms, %.2f ns per step; expired %u%s", K == 0 ? "k_phase_rows" : "k_phase_scan<64>", K, C, steps, stride,
             ms, ms * 1e6 / steps, he[0], (check && rep == 0) ? "" : "\n");
      if (check && rep == 0)
      {
        printf("; cells differing from the host's loops: %zu (first: channel %zu, cell %zu)\n", bad, first / stride, first % stride);
      }
    }
  }
  return 0;
}
