// tests/cpp/mock_hrfd.cc -- a MOCK of the libhrfd C ABI (include/hrfd.h) for the CPU-only sanitizer job of the shim
// classes (tests/test_sanitizers.py): hrfd_shim.cc + shim_demo.cc are compiled against this file under
// -fsanitize=address,undefined, so that the shim's own host logic -- buffer sizes, the heap block behind the
// layout-contained IqDataProcessor, callback hand-over, ring pacing, file playback -- runs under the sanitizers without a
// GPU.  The mock computes nothing real: every output is a deterministic function of the call (sizes are the ABI's), and
// every buffer the ABI says it fills is written in full, which is what lets AddressSanitizer see a caller's short buffer.
// The transmit ring is the REAL one (hrfd_txring.hip is host code).
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/hrfd.h"

static int fail(int code, const char *, ...) { return code; }
#include "../../hackrfdiags_amd/csrc/hrfd_txring.hip"

struct hrfd_rx { uint32_t n; int mode; };
struct hrfd_demod { uint32_t n; int mode; };
struct hrfd_mod { uint32_t n; int kind; };
struct hrfd_play { uint32_t n; std::vector<int8_t> file; std::vector<uint32_t> pos; };
struct hrfd_nco { uint32_t n; float phase; };

extern "C" {
const char *hrfd_last_error(void) { return "mock"; }
int hrfd_version(void) { return 1; }
int hrfd_device_count(void) { return 1; }

int hrfd_rx_create(uint32_t n, int, hrfd_rx **out) { *out = new hrfd_rx{n, 0}; return HRFD_OK; }
int hrfd_rx_destroy(hrfd_rx *h) { delete h; return HRFD_OK; }
int hrfd_rx_set_mode(hrfd_rx *h, uint32_t, int mode) { h->mode = mode; return HRFD_OK; }
int hrfd_rx_set_gain(hrfd_rx *, uint32_t, int, float) { return HRFD_OK; }
int hrfd_rx_set_threshold(hrfd_rx *, uint32_t, int32_t) { return HRFD_OK; }
int hrfd_rx_reset_demod(hrfd_rx *, uint32_t, int) { return HRFD_OK; }
int hrfd_rx_process_block(hrfd_rx *h, const int8_t *iq, uint32_t block_bytes, uint32_t n_blocks, uint32_t, int16_t *pcm,
                          uint32_t *n_pcm, uint32_t *magnitude, uint8_t *allowed, int8_t *iq256)
{
  const uint32_t per = block_bytes / 512;
  for (uint32_t c = 0; c < h->n; c++)
    for (uint32_t b = 0; b < n_blocks; b++)
    {
      const int8_t *x = iq + ((size_t)c * n_blocks + b) * block_bytes;
      long sum = 0;
      for (uint32_t i = 0; i < block_bytes; i++) sum += x[i];       // (reads every input byte: a short input shows)
      for (uint32_t i = 0; i < per; i++) pcm[((size_t)c * n_blocks + b) * per + i] = (int16_t)(sum + i);
      if (n_pcm) n_pcm[c * n_blocks + b] = h->mode ? per : 0;
      if (magnitude) magnitude[c * n_blocks + b] = (uint32_t)(sum & 127);
      if (allowed) allowed[c * n_blocks + b] = 1;
      if (iq256) memset(iq256 + ((size_t)c * n_blocks + b) * (block_bytes / 8), 3, block_bytes / 8);
    }
  return HRFD_OK;
}
int hrfd_rx_reduce_sample_rate(hrfd_rx *h, const int8_t *iq, uint32_t block_bytes, int8_t *iq256)
{
  for (uint32_t c = 0; c < h->n; c++)
    for (uint32_t i = 0; i < block_bytes / 8; i++) iq256[(size_t)c * (block_bytes / 8) + i] = iq[(size_t)c * block_bytes + 8 * i];
  return HRFD_OK;
}
int hrfd_rx_sync(hrfd_rx *, uint32_t *n) { if (n) *n = 0; return HRFD_OK; }
int hrfd_rx_pending_samples(hrfd_rx *, uint32_t *pending) { *pending = 0; return HRFD_OK; }
uint32_t hrfd_rx_pcm_capacity(uint32_t block_bytes) { return (block_bytes + 511u) / 512u; }
uint32_t hrfd_rx_iq256_capacity(uint32_t block_bytes) { return 2u * ((block_bytes / 2u + 7u) / 8u); }
uint32_t hrfd_demod_pcm_capacity(uint32_t bytes) { return (bytes + 63u) / 64u; }
int hrfd_rx_failed_channels(hrfd_rx *h, uint8_t *out, uint32_t n) { memset(out, 0, n); (void)h; return HRFD_OK; }

int hrfd_demod_create(int mode, uint32_t n, int, hrfd_demod **out) { *out = new hrfd_demod{n, mode}; return HRFD_OK; }
int hrfd_demod_destroy(hrfd_demod *h) { delete h; return HRFD_OK; }
int hrfd_demod_reset(hrfd_demod *, uint32_t) { return HRFD_OK; }
int hrfd_demod_set_gain(hrfd_demod *, uint32_t, float) { return HRFD_OK; }
int hrfd_demod_set_sideband(hrfd_demod *, uint32_t, int) { return HRFD_OK; }
int hrfd_demod_process(hrfd_demod *h, const int8_t *iq256, uint32_t bytes, int16_t *pcm, uint32_t *n_pcm)
{
  const uint32_t per = bytes / 64;
  for (uint32_t c = 0; c < h->n; c++)
  {
    long sum = 0;
    for (uint32_t i = 0; i < bytes; i++) sum += iq256[(size_t)c * bytes + i];
    for (uint32_t i = 0; i < per; i++) pcm[(size_t)c * per + i] = (int16_t)(sum - i);
    if (n_pcm) n_pcm[c] = per;
  }
  return HRFD_OK;
}

int hrfd_mod_create(int kind, uint32_t n, int, hrfd_mod **out) { *out = new hrfd_mod{n, kind}; return HRFD_OK; }
int hrfd_mod_destroy(hrfd_mod *h) { delete h; return HRFD_OK; }
int hrfd_mod_reset(hrfd_mod *, uint32_t) { return HRFD_OK; }
int hrfd_mod_set_sideband(hrfd_mod *, uint32_t, int) { return HRFD_OK; }
int hrfd_mod_set_modulation_index(hrfd_mod *, uint32_t, float) { return HRFD_OK; }
int hrfd_mod_set_deviation(hrfd_mod *, uint32_t, float) { return HRFD_OK; }
int hrfd_mod_process(hrfd_mod *h, const int16_t *pcm, uint32_t n_per, int8_t *iq_out, uint32_t *out_bytes)
{
  const uint32_t in_per = (h->kind == HRFD_MOD_INTERP) ? 2 * n_per : n_per;
  for (uint32_t c = 0; c < h->n; c++)
  {
    long sum = 0;
    for (uint32_t i = 0; i < in_per; i++) sum += pcm[(size_t)c * in_per + i];
    memset(iq_out + (size_t)c * 512 * n_per, (int)(sum & 63), (size_t)512 * n_per);
  }
  if (out_bytes) *out_bytes = 512 * n_per;
  return HRFD_OK;
}
int hrfd_mod_sync(hrfd_mod *) { return HRFD_OK; }

int hrfd_play_create(uint32_t n, int, hrfd_play **out) { *out = new hrfd_play{n, {}, std::vector<uint32_t>(n, 0)}; return HRFD_OK; }
int hrfd_play_destroy(hrfd_play *h) { delete h; return HRFD_OK; }
int hrfd_play_load(hrfd_play *h, const int8_t *bytes, uint32_t n) { h->file.assign(bytes, bytes + n); return HRFD_OK; }
int hrfd_play_load_file(hrfd_play *h, const char *path)
{
  FILE *f = fopen(path, "rb");
  if (!f) return HRFD_EINVAL;
  h->file.clear();
  int8_t buf[4096];
  size_t k;
  while ((k = fread(buf, 1, sizeof(buf), f)) > 0) h->file.insert(h->file.end(), buf, buf + k);
  fclose(f);
  return h->file.empty() ? HRFD_EINVAL : HRFD_OK;
}
int hrfd_play_set_position(hrfd_play *h, uint32_t c, uint32_t p) { h->pos[c % h->n] = p; return HRFD_OK; }
int hrfd_play_get_position(hrfd_play *h, uint32_t c, uint32_t *p) { *p = h->pos[c % h->n]; return HRFD_OK; }
int hrfd_play_get(hrfd_play *h, int8_t *out, uint32_t bytes)
{
  if (h->file.empty()) return HRFD_ESTATE;
  for (uint32_t c = 0; c < h->n; c++)
    for (uint32_t i = 0; i < bytes; i++)
    {
      out[(size_t)c * bytes + i] = h->file[h->pos[c]];
      h->pos[c] = (h->pos[c] + 1) % (uint32_t)h->file.size();
    }
  return HRFD_OK;
}

int hrfd_nco_create(uint32_t n, float, float, int, hrfd_nco **out) { *out = new hrfd_nco{n, 0.0f}; return HRFD_OK; }
int hrfd_nco_destroy(hrfd_nco *h) { delete h; return HRFD_OK; }
int hrfd_nco_set_frequency(hrfd_nco *, uint32_t, float) { return HRFD_OK; }
int hrfd_nco_reset(hrfd_nco *h, uint32_t) { h->phase = 0; return HRFD_OK; }
int hrfd_nco_run(hrfd_nco *h, int, uint32_t count, float *i_out, float *q_out)
{
  for (uint32_t c = 0; c < h->n; c++)
    for (uint32_t k = 0; k < count; k++) { i_out[(size_t)c * count + k] = 1.0f; q_out[(size_t)c * count + k] = 0.0f; }
  return HRFD_OK;
}
}
