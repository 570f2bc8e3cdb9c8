// k_phase_scan alone on synthetic steps: ms per launch, ns per step, and (built with -DHRFD_PS_PROBE) the share of
// the recurrence wave's cycles spent waiting for the loaders.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DHRFD_PS_PROBE -I../../hackrfdiags_amd/csrc -o phase_scan_rate phase_scan_rate.hip
#include "hrfd_rx_kernels.hip"
#include "../../include/hrfd.h"
#include "hrfd_tx_kernels.hip"
#include <stdio.h>
#include <vector>
using namespace hrfd;
#ifndef HRFD_PS_CHAN
#define HRFD_PS_CHAN 64
#endif
constexpr int kPsChan = HRFD_PS_CHAN;

int main(int argc, char **argv)
{
  const uint32_t C = argc > 1 ? atoi(argv[1]) : 1024;
  const size_t steps = argc > 2 ? atol(argv[2]) : 262144;
  uint32_t *cells, *err;
  float *acc;
  hipMalloc(&cells, (size_t)C * steps * 4);
  hipMalloc(&acc, C * 4);
  hipMalloc(&err, 64);
  std::vector<float> h((size_t)C * steps);
  unsigned s = 12345;
  for (size_t i = 0; i < h.size(); i++)
  {
    s = s * 1664525u + 1013904223u;
    h[i] = ((int)(s >> 8) % 2000 - 1000) * 1.8e-3f;
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 4; rep++)
  {
    hipMemcpy(cells, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(acc, 0, C * 4);
    hipMemset(err, 0, 64);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_phase_scan<kPsChan>, dim3((C + kPsChan - 1) / kPsChan), dim3(kPsThreads), 0, 0, cells, steps, acc, C, err);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    uint32_t he[4];
    hipMemcpy(he, err, 16, hipMemcpyDeviceToHost);
    const double wgs = (C + kPsChan - 1) / kPsChan;
    printf("%u channels x %zu steps: %.3f ms, %.2f ns per step; expired %u; recurrence wave: %.0f cycles per step, %.1f %% of them waiting for chunks\n",
           C, steps, ms, ms * 1e6 / steps, he[0], he[1] * 256.0 / wgs / steps, he[1] ? 100.0 * he[2] / he[1] : 0.0);
  }
  return 0;
}
