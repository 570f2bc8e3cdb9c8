// hackrfdiags_amd/csrc/hrfd_play.hip -- hrfd_play_*: cyclic playback of an .iq file from HBM.
//
// DataProvider (src_diags/DataProvider.cc) holds the file in a host buffer and
// retrieveIqDataFromBuffer (:174-231) memcpy's the next byteCount bytes out of it, wrapping at the
// end, iqSampleBufferIndex %= iqSampleBufferLength.  Here the image lives in device memory once,
// every channel has its own read position, and one launch fills [C][bytes] -- the source of soak
// tests of the receive path.  Pure HBM copy: coalesced 16-byte stores, the
// unaligned source read as two dwords + v_alignbyte.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>

namespace hrfd {

struct PlayParams
{
  const uint8_t *ring;      // file image, padded with 8 readable bytes
  uint32_t length;          // iqSampleBufferLength
  const uint32_t *index;    // [C] iqSampleBufferIndex
  int8_t *out;
  uint64_t ch_stride;
  uint32_t bytes;           // per channel
  uint32_t n_channels;
};

// one thread = 16 output bytes
__global__ __launch_bounds__(256) void k_play(const PlayParams Q)
{
  const uint32_t per = (Q.bytes + 15u) / 16u;
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (uint64_t)per * Q.n_channels)
  {
    return;
  }
  const uint32_t c = (uint32_t)(t / per);
  const uint32_t o = (uint32_t)(t - (uint64_t)c * per) * 16u;          // byte offset in the channel's output
  uint32_t src = (uint32_t)(((uint64_t)Q.index[c] + o) % Q.length);
  int8_t *dst = Q.out + (uint64_t)c * Q.ch_stride + o;
  const uint32_t nb = min(16u, Q.bytes - o);
  if (nb == 16u && src + 16u <= Q.length && (Q.ch_stride & 15u) == 0u && ((uintptr_t)Q.out & 15u) == 0u)
  {
    // fast path: no wrap inside these 16 bytes
    const uint32_t *w = reinterpret_cast<const uint32_t *>(Q.ring + (src & ~3u));
    const uint32_t sh = src & 3u;
    uint32_t a[5];
#pragma unroll
    for (int k = 0; k < 5; k++)
    {
      a[k] = w[k];                                         // the image is padded: reading one dword past is safe
    }
    uint4 v;
    v.x = __builtin_amdgcn_alignbyte(a[1], a[0], sh);
    v.y = __builtin_amdgcn_alignbyte(a[2], a[1], sh);
    v.z = __builtin_amdgcn_alignbyte(a[3], a[2], sh);
    v.w = __builtin_amdgcn_alignbyte(a[4], a[3], sh);
    *reinterpret_cast<uint4 *>(dst) = v;
    return;
  }
  for (uint32_t k = 0; k < nb; k++)
  {
    dst[k] = (int8_t)Q.ring[src];
    src = (src + 1u == Q.length) ? 0u : src + 1u;
  }
}

} // namespace hrfd

struct hrfd_play
{
  int device = 0;
  uint32_t n_channels = 0;
  hipStream_t stream = nullptr;
  uint8_t *d_ring = nullptr;
  uint32_t length = 0;                 // 0: nothing loaded
  std::vector<uint32_t> index;         // host mirror of the read positions
  uint32_t *d_index = nullptr;
  int8_t *d_out = nullptr;             // staging of hrfd_play_get
  size_t cap_out = 0;
};

extern "C" int hrfd_play_create(uint32_t n_channels, int device, hrfd_play **out)
{
  using namespace hrfd;
  if (out == nullptr || n_channels == 0)
  {
    return fail(HRFD_EINVAL, "hrfd_play_create: n_channels > 0 and a result pointer");
  }
  *out = nullptr;
  if (hrfd_device_count() <= 0)
  {
    return fail(HRFD_ENODEV, "hrfd_play_create: no HIP device visible (this library has no CPU path)");
  }
  if (device < 0)
  {
    HIP_TRY(hipGetDevice(&device));
  }
  HIP_TRY(hipSetDevice(device));
  int rc = HRFD_OK;
  hrfd_play *h = new hrfd_play;
  h->device = device;
  h->n_channels = n_channels;
  h->index.assign(n_channels, 0u);
  hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc((void **)&h->d_index, sizeof(uint32_t) * n_channels);
  if (e != hipSuccess)
  {
    rc = fail(HRFD_ENOMEM, "hrfd_play_create: %s", hipGetErrorString(e));
    hrfd_play_destroy(h);
    return rc;
  }
  *out = h;
  return HRFD_OK;
}

extern "C" int hrfd_play_destroy(hrfd_play *h)
{
  if (h == nullptr)
  {
    return HRFD_OK;
  }
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->d_ring) (void)hipFree(h->d_ring);
  if (h->d_index) (void)hipFree(h->d_index);
  if (h->d_out) (void)hipFree(h->d_out);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return HRFD_OK;
}

// DataProvider::loadIqFile (:235-300): a new image replaces the old one, every position restarts at 0
extern "C" int hrfd_play_load(hrfd_play *h, const int8_t *bytes, uint32_t n_bytes)
{
  using namespace hrfd;
  if (h == nullptr || bytes == nullptr || n_bytes == 0)
  {
    return fail(HRFD_EINVAL, "hrfd_play_load: NULL argument or empty image");
  }
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->d_ring)
  {
    (void)hipFree(h->d_ring);
    h->d_ring = nullptr;
    h->length = 0;
  }
  hipError_t e = hipMalloc((void **)&h->d_ring, (size_t)n_bytes + 8);
  if (e != hipSuccess)
  {
    return fail(HRFD_ENOMEM, "hrfd_play_load: hipMalloc(%u): %s", n_bytes, hipGetErrorString(e));
  }
  HIP_TRY(hipMemset(h->d_ring + n_bytes, 0, 8));
  HIP_TRY(hipMemcpy(h->d_ring, bytes, n_bytes, hipMemcpyHostToDevice));
  h->length = n_bytes;
  h->index.assign(h->n_channels, 0u);
  return HRFD_OK;
}

extern "C" int hrfd_play_load_file(hrfd_play *h, const char *path)
{
  using namespace hrfd;
  if (h == nullptr || path == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_play_load_file: NULL argument");
  }
  FILE *f = fopen(path, "r");
  if (f == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_play_load_file: cannot open %s", path);
  }
  fseek(f, 0L, SEEK_END);
  const long len = ftell(f);
  fseek(f, 0L, SEEK_SET);
  if (len <= 0 || len > 0x7fffffffL)
  {
    fclose(f);
    return fail(HRFD_EINVAL, "hrfd_play_load_file: %s is empty or larger than 2 GiB", path);
  }
  std::vector<int8_t> buf((size_t)len);
  const size_t got = fread(buf.data(), 1, (size_t)len, f);
  fclose(f);
  if (got != (size_t)len)
  {
    return fail(HRFD_EINVAL, "hrfd_play_load_file: short read of %s", path);
  }
  return hrfd_play_load(h, buf.data(), (uint32_t)len);
}

extern "C" int hrfd_play_set_position(hrfd_play *h, uint32_t channel, uint32_t byte_index)
{
  using namespace hrfd;
  if (h == nullptr || (channel != HRFD_ALL_CHANNELS && channel >= h->n_channels) || h->length == 0 ||
      byte_index >= h->length)
  {
    return fail(HRFD_EINVAL, "hrfd_play_set_position: bad handle, channel or index (a file must be loaded)");
  }
  for (uint32_t c = 0; c < h->n_channels; c++)
  {
    if (channel == HRFD_ALL_CHANNELS || channel == c)
    {
      h->index[c] = byte_index;
    }
  }
  return HRFD_OK;
}

extern "C" int hrfd_play_get_position(hrfd_play *h, uint32_t channel, uint32_t *byte_index)
{
  using namespace hrfd;
  if (h == nullptr || channel >= h->n_channels || byte_index == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_play_get_position: bad argument");
  }
  *byte_index = h->index[channel];
  return HRFD_OK;
}

extern "C" int hrfd_play_get_device(hrfd_play *h, int8_t *d_out, uint64_t channel_stride,
                                    uint32_t bytes_per_channel, void *stream)
{
  using namespace hrfd;
  if (h == nullptr || d_out == nullptr || channel_stride < bytes_per_channel)
  {
    return fail(HRFD_EINVAL, "hrfd_play_get_device: NULL argument or channel_stride < bytes_per_channel");
  }
  if (h->length == 0 || bytes_per_channel == 0)
  {
    return HRFD_OK;                                        // DataProvider.cc:181: no file, no action
  }
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = (stream != nullptr) ? (hipStream_t)stream : h->stream;
  HIP_TRY(hipMemcpyAsync(h->d_index, h->index.data(), sizeof(uint32_t) * h->n_channels, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));                        // the host vector changes below
  PlayParams Q;
  Q.ring = h->d_ring;
  Q.length = h->length;
  Q.index = h->d_index;
  Q.out = d_out;
  Q.ch_stride = channel_stride;
  Q.bytes = bytes_per_channel;
  Q.n_channels = h->n_channels;
  const uint64_t threads = (uint64_t)((bytes_per_channel + 15u) / 16u) * h->n_channels;
  hipLaunchKernelGGL(k_play, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, s, Q);
  HIP_TRY(hipGetLastError());
  for (uint32_t c = 0; c < h->n_channels; c++)
  {
    h->index[c] = (uint32_t)(((uint64_t)h->index[c] + bytes_per_channel) % h->length);
  }
  return HRFD_OK;
}

extern "C" int hrfd_play_get(hrfd_play *h, int8_t *out, uint32_t bytes_per_channel)
{
  using namespace hrfd;
  if (h == nullptr || out == nullptr)
  {
    return fail(HRFD_EINVAL, "hrfd_play_get: NULL argument");
  }
  if (h->length == 0 || bytes_per_channel == 0)
  {
    return HRFD_OK;
  }
  HIP_TRY(hipSetDevice(h->device));
  const uint64_t stride = ((uint64_t)bytes_per_channel + 15u) & ~(uint64_t)15u;
  const size_t need = (size_t)stride * h->n_channels;
  if (need > h->cap_out)
  {
    HIP_TRY(hipStreamSynchronize(h->stream));
    int rc = grow((void **)&h->d_out, &h->cap_out, need);
    if (rc != HRFD_OK) return rc;
  }
  int rc = hrfd_play_get_device(h, h->d_out, stride, bytes_per_channel, h->stream);
  if (rc != HRFD_OK)
  {
    return rc;
  }
  HIP_TRY(hipMemcpy2DAsync(out, bytes_per_channel, h->d_out, stride, bytes_per_channel, h->n_channels,
                           hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return HRFD_OK;
}
