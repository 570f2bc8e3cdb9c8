// hrfd_membw.hip -- two plain stream kernels, the denominators bench.py quotes beside the HBM spec peak: what a kernel
// that does NOTHING but read (or nothing but write) a buffer reaches on this GPU in this run ("also report
// against a measured stream bandwidth from the same run").  Not part of the drop-in boundary (include/hrfd_debug.h).
//   read : one workgroup of 256 lanes per 32 KiB chunk, eight 16-byte loads per lane in flight, every XCD (workgroup
//          ids go round the eight XCDs, each with an L2 of its own) reading one contiguous eighth of the buffer;
//   write: the same shape storing (tools/ubench/store_shapes.hip: 6.4 TB/s in this shape, 5.7 when consecutive chunks
//          go round the XCDs).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hrfd {

constexpr int kBwThreads = 256, kBwRounds = 8;             // 256 x 8 x 16 B = 32 KiB per workgroup

__device__ __forceinline__ size_t bw_chunk(size_t chunks)
{
  const size_t per = chunks / 8;
  const uint32_t x = blockIdx.x & 7u, i = blockIdx.x >> 3;
  return (size_t)x * per + i;
}

__global__ __launch_bounds__(kBwThreads) void k_membw_read(const uint4 *__restrict__ in, size_t chunks, uint32_t *sink)
{
  const uint4 *p = in + bw_chunk(chunks) * (size_t)(kBwRounds * kBwThreads) + threadIdx.x;
  uint4 q[kBwRounds];
#pragma unroll
  for (int r = 0; r < kBwRounds; r++)
  {
    q[r] = p[r * kBwThreads];
  }
  uint32_t acc = 0;
#pragma unroll
  for (int r = 0; r < kBwRounds; r++)
  {
    acc ^= q[r].x ^ q[r].y ^ q[r].z ^ q[r].w;
  }
  if (acc == 0x9e3779b9u && sink != nullptr)              // (never for the buffers bench.py reads; keeps the loads alive)
  {
    sink[0] = acc;
  }
}

__global__ __launch_bounds__(kBwThreads) void k_membw_write(uint4 *out, size_t chunks)
{
  const size_t ch = bw_chunk(chunks);
  uint4 *o = out + ch * (size_t)(kBwRounds * kBwThreads) + threadIdx.x;
#pragma unroll
  for (int r = 0; r < kBwRounds; r++)
  {
    o[r * kBwThreads] = make_uint4((uint32_t)ch, (uint32_t)r, threadIdx.x, 7u);
  }
}

}  // namespace hrfd

// kind 0: read `bytes` of d_buf, kind 1: overwrite them.  bytes: a multiple of 256 KiB (eight XCDs x 32 KiB chunks).
extern "C" int hrfd_debug_membw(int kind, void *d_buf, size_t bytes, void *d_sink, void *stream)
{
  if (d_buf == nullptr || bytes == 0 || (bytes % (8u * 32768u)) != 0 || (kind != 0 && kind != 1) || bytes / 32768u > 0x7fffffffu)
  {
    return fail(HRFD_EINVAL, "hrfd_debug_membw: kind 0 | 1, bytes a multiple of 256 KiB");
  }
  const size_t chunks = bytes / 32768u;
  if (kind == 0)
  {
    hipLaunchKernelGGL(hrfd::k_membw_read, dim3((uint32_t)chunks), dim3(hrfd::kBwThreads), 0, (hipStream_t)stream,
                       (const uint4 *)d_buf, chunks, (uint32_t *)d_sink);
  }
  else
  {
    hipLaunchKernelGGL(hrfd::k_membw_write, dim3((uint32_t)chunks), dim3(hrfd::kBwThreads), 0, (hipStream_t)stream, (uint4 *)d_buf, chunks);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess)
  {
    return fail(HRFD_ENODEV, "hrfd_debug_membw: %s", hipGetErrorString(e));
  }
  return HRFD_OK;
}
