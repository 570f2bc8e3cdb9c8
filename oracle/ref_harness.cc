// oracle/ref_harness.cc -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A thin extern "C" driver around the *unmodified* reference sources of
// wizardyesterday/HackRfDiags, compiled where they lie under /root/reference by
// oracle/Makefile into oracle/_ref/libhrfd_ref.so (git-ignored; never committed;
// no reference source is copied into this repository).  It exists so that
//   * oracle/hrfd_oracle.c (our own CPU restatement) can be pinned against the
//     real reference on arbitrary inputs (tests/test_oracle_vs_ref.py),
//   * tests/golden/make_golden.py can generate golden vectors, and
//   * bench.py can time the reference's own CPU chain as `cpu_baseline`
//     (kind "reference").
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// the resulting library.
//
// The reference classes driven here:
//   IqDataProcessor::acceptIqData       radioDiags/src_diags/IqDataProcessor.cc:926
//   {Am,Fm,WbFm,Ssb}Demodulator::acceptIqData  radioDiags/*Demodulator/*.cc
//   SsbModulator::acceptData            radioDiags/SsbModulator/SsbModulator.cc:455
//   Nco::run / Nco::runFast             radioDiags/Nco/Nco.cc:186,222
//   Decimator_int16 / FirFilter_int16 / Interpolator_int16 / FirFilter / IirFilter
//
// The reference links against two symbols of its host application; we supply
// them exactly as SURVEY.md section 8(c) describes.

#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

// The harness (and only the harness) peeks at private members to dump
// intermediate streams (decimatedData, demodulatedData).  The reference
// translation units themselves are compiled untouched.
#define private public
#include "IqDataProcessor.h"
#include "SsbModulator.h"
#include "AmModulator.h"
#include "FmModulator.h"
#include "WbFmModulator.h"
#include "BasebandDataProcessor.h"
#include "UdpClient.h"
#include "DataProvider.h"
#include "Nco.h"
#include "Interpolator_int16.h"
#include "FirFilter_int16.h"
#include "FirFilter.h"
#include "IirFilter.h"
#include "DbfsCalculator.h"
#undef private

// Symbols the reference expects from radioDiags (Radio.cc:15, diagUi.cc:2881).
uint32_t radio_adjustableReceiveGainInDb = 0;
void nprintf(FILE *, const char *, ...) {}

namespace {

struct PcmSink
{
  std::vector<int16_t> pcm;
  uint32_t callbacks;
};

// The reference's PCM callback carries no context pointer, so the harness
// routes it through a "current sink" that is set around every call.
thread_local PcmSink *g_sink = nullptr;

void pcmCallback(int16_t *bufferPtr, uint32_t bufferLength)
{
  if (g_sink != nullptr)
  {
    g_sink->pcm.insert(g_sink->pcm.end(), bufferPtr, bufferPtr + bufferLength);
    g_sink->callbacks++;
  }
}

struct RefRx
{
  IqDataProcessor *proc;
  AmDemodulator *am;
  FmDemodulator *fm;
  WbFmDemodulator *wbfm;
  SsbDemodulator *ssb;
  PcmSink sink;
};

struct RefDemod
{
  int mode;
  AmDemodulator *am;
  FmDemodulator *fm;
  WbFmDemodulator *wbfm;
  SsbDemodulator *ssb;
  PcmSink sink;
};

uint32_t drain(PcmSink &sink, int16_t *pcmOut, uint32_t pcmCapacity)
{
  uint32_t n = (uint32_t)sink.pcm.size();
  if (n > pcmCapacity)
  {
    n = pcmCapacity;
  }
  if (n > 0 && pcmOut != nullptr)
  {
    memcpy(pcmOut, sink.pcm.data(), n * sizeof(int16_t));
  }
  sink.pcm.clear();
  return n;
}

} // namespace

extern "C" {

//---------------------------------------------------------------- rx, outer
void *ref_rx_create(void)
{
  static char ip[] = "127.0.0.1";
  RefRx *h = new RefRx;
  h->sink.callbacks = 0;
  h->proc = new IqDataProcessor(ip, 8001);
  h->am = new AmDemodulator(pcmCallback);
  h->fm = new FmDemodulator(pcmCallback);
  h->wbfm = new WbFmDemodulator(pcmCallback);
  h->ssb = new SsbDemodulator(pcmCallback);
  // Same wiring as Radio.cc:179-203.
  h->proc->setAmDemodulator(h->am);
  h->proc->setFmDemodulator(h->fm);
  h->proc->setWbFmDemodulator(h->wbfm);
  h->proc->setSsbDemodulator(h->ssb);
  h->proc->disableIqDump();
  return h;
}

void ref_rx_destroy(void *hv)
{
  RefRx *h = (RefRx *)hv;
  delete h->proc;
  delete h->am;
  delete h->fm;
  delete h->wbfm;
  delete h->ssb;
  delete h;
}

// mode: IqDataProcessor::demodulatorType {None=0,Am=1,Fm=2,WbFm=3,Lsb=4,Usb=5}
void ref_rx_set_mode(void *hv, int mode)
{
  RefRx *h = (RefRx *)hv;
  h->proc->setDemodulatorMode((IqDataProcessor::demodulatorType)mode);
}

void ref_rx_set_gain(void *hv, int mode, float gain)
{
  RefRx *h = (RefRx *)hv;
  switch (mode)
  {
    case 1: h->am->setDemodulatorGain(gain); break;
    case 2: h->fm->setDemodulatorGain(gain); break;
    case 3: h->wbfm->setDemodulatorGain(gain); break;
    case 4:
    case 5: h->ssb->setDemodulatorGain(gain); break;
    default: break;
  }
}

void ref_rx_set_threshold(void *hv, int32_t threshold)
{
  RefRx *h = (RefRx *)hv;
  h->proc->setSignalDetectThreshold(threshold);
}

void ref_set_receive_gain_db(uint32_t gainInDb)
{
  radio_adjustableReceiveGainInDb = gainInDb;
}

// One IqDataProcessor::acceptIqData call.  Returns the number of PCM samples
// the demodulator handed to the callback (0 when squelched or mode None).
// iq256Out (optional, >= 2 * ceil(byteCount / 16) bytes: what a call can complete at most) receives decimatedData
// after the Fs/4 mix -- the call's own count is 2 * floor((held + byteCount / 2) / 8), ref_rx_reduce_sample_rate
// returns it for the front end alone; magnitudeOut (optional) the squelch's block-mean magnitude.
uint32_t ref_rx_process(void *hv,
                        const int8_t *bufferPtr,
                        uint32_t byteCount,
                        int16_t *pcmOut,
                        uint32_t pcmCapacity,
                        uint32_t *magnitudeOut,
                        int8_t *iq256Out)
{
  RefRx *h = (RefRx *)hv;
  std::vector<int8_t> scratch(bufferPtr, bufferPtr + byteCount);
  g_sink = &h->sink;
  h->proc->acceptIqData(0, scratch.data(), byteCount);
  g_sink = nullptr;
  if (magnitudeOut != nullptr)
  {
    *magnitudeOut = h->proc->squelchPtr->getSignalMagnitude();
  }
  if (iq256Out != nullptr)
  {
    memcpy(iq256Out, h->proc->decimatedData, 2u * ((byteCount / 2u + 7u) / 8u));
  }
  return drain(h->sink, pcmOut, pcmCapacity);
}

// IqDataProcessor::reduceSampleRate (IqDataProcessor.cc:429-500, public): the front end alone.  Returns its byteCount
// and copies that many bytes of decimatedData (no Fs/4 mix).
uint32_t ref_rx_reduce_sample_rate(void *hv, const int8_t *bufferPtr, uint32_t byteCount, int8_t *iq256Out)
{
  RefRx *h = (RefRx *)hv;
  std::vector<int8_t> scratch(bufferPtr, bufferPtr + byteCount);
  scratch.push_back(0);                                    // (an odd count reads one byte past the end, :474)
  const uint32_t n = h->proc->reduceSampleRate(scratch.data(), byteCount);
  if (iq256Out != nullptr)
  {
    memcpy(iq256Out, h->proc->decimatedData, n);
  }
  return n;
}

// WBFM only: copy the float stream produced by the last demodulateSignal call.
void ref_rx_wbfm_float_stream(void *hv, float *out, uint32_t count)
{
  RefRx *h = (RefRx *)hv;
  memcpy(out, h->wbfm->demodulatedData, count * sizeof(float));
}

// bench.py's cpu_baseline leg: `threads` host threads, each with an IqDataProcessor
// and its four demodulators of its own (the wiring of Radio.cc:179-203), looping over the same
// n_blocks x block_bytes of IQ for `seconds` the way DataConsumer's consumer thread does
// (DataConsumer.cc:319-351: one acceptIqData call per 262144-byte block).  The loop allocates
// nothing: every thread copies the input once (acceptIqData takes a non-const buffer) and the PCM
// sink keeps its capacity.  Returns the number of blocks all threads processed and the wall time
// from the common start to the last thread's end.
uint64_t ref_bench_rx(int mode, uint32_t threads, double seconds, const int8_t *iq, uint32_t n_blocks,
                      uint32_t block_bytes, double *elapsed_s, uint64_t *pcm_samples)
{
  if (threads == 0 || n_blocks == 0)
  {
    return 0;
  }
  std::vector<RefRx *> rx(threads);
  for (uint32_t t = 0; t < threads; t++)
  {
    rx[t] = (RefRx *)ref_rx_create();
    ref_rx_set_mode(rx[t], mode);
    rx[t]->sink.pcm.reserve(4096);
  }
  std::vector<uint64_t> blocks(threads, 0), samples(threads, 0);
  std::vector<double> ends(threads, 0.0);
  std::atomic<uint32_t> ready(0);
  std::atomic<bool> go(false);
  typedef std::chrono::steady_clock clk;
  clk::time_point t0;
  auto body = [&](uint32_t t) {
    std::vector<int8_t> mine(iq, iq + (size_t)n_blocks * block_bytes);
    RefRx *h = rx[t];
    g_sink = &h->sink;
    ready.fetch_add(1);
    while (!go.load(std::memory_order_acquire))
    {
      std::this_thread::yield();
    }
    uint64_t n = 0, pcm = 0;
    for (;;)
    {
      for (uint32_t b = 0; b < n_blocks; b++)
      {
        h->proc->acceptIqData(0, mine.data() + (size_t)b * block_bytes, block_bytes);
        pcm += h->sink.pcm.size();
        h->sink.pcm.clear();
      }
      n += n_blocks;
      if (std::chrono::duration<double>(clk::now() - t0).count() >= seconds)
      {
        break;
      }
    }
    g_sink = nullptr;
    ends[t] = std::chrono::duration<double>(clk::now() - t0).count();
    blocks[t] = n;
    samples[t] = pcm;
  };
  std::vector<std::thread> pool;
  for (uint32_t t = 0; t < threads; t++)
  {
    pool.emplace_back(body, t);
  }
  while (ready.load() < threads)
  {
    std::this_thread::yield();
  }
  t0 = clk::now();
  go.store(true, std::memory_order_release);
  for (std::thread &th : pool)
  {
    th.join();
  }
  uint64_t total = 0, pcm_total = 0;
  double last = 0.0;
  for (uint32_t t = 0; t < threads; t++)
  {
    total += blocks[t];
    pcm_total += samples[t];
    last = ends[t] > last ? ends[t] : last;
    ref_rx_destroy(rx[t]);
  }
  if (elapsed_s != nullptr) *elapsed_s = last;
  if (pcm_samples != nullptr) *pcm_samples = pcm_total;
  return total;
}

//---------------------------------------------------------------- rx, inner
void *ref_demod_create(int mode)
{
  RefDemod *h = new RefDemod;
  h->mode = mode;
  h->sink.callbacks = 0;
  h->am = nullptr; h->fm = nullptr; h->wbfm = nullptr; h->ssb = nullptr;
  switch (mode)
  {
    case 1: h->am = new AmDemodulator(pcmCallback); break;
    case 2: h->fm = new FmDemodulator(pcmCallback); break;
    case 3: h->wbfm = new WbFmDemodulator(pcmCallback); break;
    case 4: h->ssb = new SsbDemodulator(pcmCallback);
            h->ssb->setLsbDemodulationMode(); break;
    case 5: h->ssb = new SsbDemodulator(pcmCallback);
            h->ssb->setUsbDemodulationMode(); break;
    default: break;
  }
  return h;
}

void ref_demod_destroy(void *hv)
{
  RefDemod *h = (RefDemod *)hv;
  delete h->am; delete h->fm; delete h->wbfm; delete h->ssb;
  delete h;
}

void ref_demod_set_gain(void *hv, float gain)
{
  RefDemod *h = (RefDemod *)hv;
  if (h->am) h->am->setDemodulatorGain(gain);
  if (h->fm) h->fm->setDemodulatorGain(gain);
  if (h->wbfm) h->wbfm->setDemodulatorGain(gain);
  if (h->ssb) h->ssb->setDemodulatorGain(gain);
}

void ref_demod_reset(void *hv)
{
  RefDemod *h = (RefDemod *)hv;
  if (h->am) h->am->resetDemodulator();
  if (h->fm) h->fm->resetDemodulator();
  if (h->wbfm) h->wbfm->resetDemodulator();
  if (h->ssb) h->ssb->resetDemodulator();
}

void ref_demod_set_sideband(void *hv, int lsb)
{
  RefDemod *h = (RefDemod *)hv;
  if (h->ssb)
  {
    if (lsb) h->ssb->setLsbDemodulationMode();
    else h->ssb->setUsbDemodulationMode();
  }
}

// X::acceptIqData(int8_t*,uint32_t) on 256 kS/s, already mixed, IQ (<= 32768 B).
uint32_t ref_demod_process(void *hv,
                           const int8_t *bufferPtr,
                           uint32_t byteCount,
                           int16_t *pcmOut,
                           uint32_t pcmCapacity)
{
  RefDemod *h = (RefDemod *)hv;
  std::vector<int8_t> scratch(bufferPtr, bufferPtr + byteCount);
  g_sink = &h->sink;
  if (h->am) h->am->acceptIqData(scratch.data(), byteCount);
  if (h->fm) h->fm->acceptIqData(scratch.data(), byteCount);
  if (h->wbfm) h->wbfm->acceptIqData(scratch.data(), byteCount);
  if (h->ssb) h->ssb->acceptIqData(scratch.data(), byteCount);
  g_sink = nullptr;
  return drain(h->sink, pcmOut, pcmCapacity);
}

//---------------------------------------------------------------- tx
void *ref_ssbmod_create(int lsb)
{
  SsbModulator *m = new SsbModulator();
  if (lsb) m->setLsbModulationMode();
  else m->setUsbModulationMode();
  return m;
}

void ref_ssbmod_destroy(void *hv)
{
  delete (SsbModulator *)hv;
}

void ref_ssbmod_set_sideband(void *hv, int lsb)
{
  SsbModulator *m = (SsbModulator *)hv;
  if (lsb) m->setLsbModulationMode();
  else m->setUsbModulationMode();
}

void ref_ssbmod_reset(void *hv)
{
  ((SsbModulator *)hv)->resetModulator();
}

// SsbModulator::acceptData; sampleCount <= 512 (fixed member arrays).
uint32_t ref_ssbmod_process(void *hv,
                            const int16_t *pcmPtr,
                            uint32_t sampleCount,
                            int8_t *iqOut)
{
  SsbModulator *m = (SsbModulator *)hv;
  std::vector<int16_t> scratch(pcmPtr, pcmPtr + sampleCount);
  uint32_t outBytes = 0;
  m->acceptData(scratch.data(), sampleCount, iqOut, &outBytes);
  return outBytes;
}

// AmModulator::acceptData (AmModulator.cc:381-395); sampleCount <= 512.
void *ref_ammod_create(void) { return new AmModulator(); }
void ref_ammod_destroy(void *hv) { delete (AmModulator *)hv; }
void ref_ammod_reset(void *hv) { ((AmModulator *)hv)->resetModulator(); }
void ref_ammod_set_index(void *hv, float index) { ((AmModulator *)hv)->setModulationIndex(index); }
uint32_t ref_ammod_process(void *hv, const int16_t *pcmPtr, uint32_t sampleCount, int8_t *iqOut)
{
  std::vector<int16_t> scratch(pcmPtr, pcmPtr + sampleCount);
  uint32_t outBytes = 0;
  ((AmModulator *)hv)->acceptData(scratch.data(), sampleCount, iqOut, &outBytes);
  return outBytes;
}

// FmModulator::acceptData (FmModulator.cc:393-407); sampleCount <= 512.
void *ref_fmmod_create(void) { return new FmModulator(); }
void ref_fmmod_destroy(void *hv) { delete (FmModulator *)hv; }
void ref_fmmod_reset(void *hv) { ((FmModulator *)hv)->resetModulator(); }
void ref_fmmod_set_deviation(void *hv, float deviation) { ((FmModulator *)hv)->setFrequencyDeviation(deviation); }
uint32_t ref_fmmod_process(void *hv, const int16_t *pcmPtr, uint32_t sampleCount, int8_t *iqOut)
{
  std::vector<int16_t> scratch(pcmPtr, pcmPtr + sampleCount);
  uint32_t outBytes = 0;
  ((FmModulator *)hv)->acceptData(scratch.data(), sampleCount, iqOut, &outBytes);
  return outBytes;
}

// WbFmModulator::acceptData (WbFmModulator.cc:341-356); sampleCount <= 512.
void *ref_wbfmmod_create(void) { return new WbFmModulator(); }
void ref_wbfmmod_destroy(void *hv) { delete (WbFmModulator *)hv; }
void ref_wbfmmod_reset(void *hv) { ((WbFmModulator *)hv)->resetModulator(); }
void ref_wbfmmod_set_deviation(void *hv, float deviation) { ((WbFmModulator *)hv)->setFrequencyDeviation(deviation); }
uint32_t ref_wbfmmod_process(void *hv, const int16_t *pcmPtr, uint32_t sampleCount, int8_t *iqOut)
{
  std::vector<int16_t> scratch(pcmPtr, pcmPtr + sampleCount);
  uint32_t outBytes = 0;
  ((WbFmModulator *)hv)->acceptData(scratch.data(), sampleCount, iqOut, &outBytes);
  return outBytes;
}

// BasebandDataProcessor's PCM ring (BasebandDataProcessor.cc:410-425,476-606), driven without its
// reader thread: the state start()/stop() set is written directly (members reached through the
// harness's `#define private public`).
void *ref_txring_create(void) { return new BasebandDataProcessor(); }
void ref_txring_destroy(void *hv)
{
  // the destructor calls stop(), which joins the reader thread when Running: there is none here
  ((BasebandDataProcessor *)hv)->streamState = BasebandDataProcessor::Idle;
  delete (BasebandDataProcessor *)hv;
}
void ref_txring_set_running(void *hv, int running)
{
  BasebandDataProcessor *p = (BasebandDataProcessor *)hv;
  if (running)
  {
    p->streamState = BasebandDataProcessor::Running;
  }
  else if (p->streamState == BasebandDataProcessor::Running)
  {
    p->streamState = BasebandDataProcessor::Idle;
    p->synchronized = false;
  }
}
void ref_txring_write(void *hv, const int16_t *pcm512)
{
  int16_t *dst = ((BasebandDataProcessor *)hv)->getNextUnfilledBuffer();
  memcpy(dst, pcm512, 512 * sizeof(int16_t));
}
void ref_txring_read(void *hv, int16_t *pcm512)
{
  const int16_t *src = ((BasebandDataProcessor *)hv)->getNextFilledBuffer();
  memcpy(pcm512, src, 512 * sizeof(int16_t));
}
void ref_txring_stats(void *hv, uint32_t *out6)
{
  BasebandDataProcessor *p = (BasebandDataProcessor *)hv;
  out6[0] = p->buffersProduced; out6[1] = p->buffersConsumed;
  out6[2] = p->pcmBlocksDropped; out6[3] = p->pcmBlocksAdded;
  out6[4] = p->pcmWriterIndex; out6[5] = p->pcmReaderIndex;
}

//---------------------------------------------------------------- Nco
void *ref_nco_create(float sampleRate, float frequency)
{
  return new Nco(sampleRate, frequency);
}

void ref_nco_destroy(void *hv)
{
  delete (Nco *)hv;
}

void ref_nco_set_frequency(void *hv, float frequency)
{
  ((Nco *)hv)->setFrequency(frequency);
}

void ref_nco_reset(void *hv)
{
  ((Nco *)hv)->reset();
}

void ref_nco_run(void *hv, int fast, uint32_t count, float *iOut, float *qOut)
{
  Nco *n = (Nco *)hv;
  for (uint32_t k = 0; k < count; k++)
  {
    if (fast) n->runFast(&iOut[k], &qOut[k]);
    else n->run(&iOut[k], &qOut[k]);
  }
}

void ref_nco_tables(void *hv, float *sinOut, float *cosOut)
{
  Nco *n = (Nco *)hv;
  memcpy(sinOut, n->Sin, sizeof(n->Sin));
  memcpy(cosOut, n->Cos, sizeof(n->Cos));
}

//---------------------------------------------------------------- primitives
// Quantised coefficients exactly as the reference constructors produce them.
void ref_quantise(const float *coefficientsPtr, int count, int16_t *out)
{
  std::vector<float> c(coefficientsPtr, coefficientsPtr + count);
  FirFilter_int16 f(count, c.data());
  memcpy(out, f.coefficientStoragePtr, count * sizeof(int16_t));
}

// Decimator_int16 over a buffer; returns number of outputs.
uint32_t ref_decimate(const float *coefficientsPtr, int taps, int factor,
                      const int16_t *in, uint32_t count, int16_t *out)
{
  std::vector<float> c(coefficientsPtr, coefficientsPtr + taps);
  Decimator_int16 d(taps, c.data(), factor);
  uint32_t n = 0;
  for (uint32_t k = 0; k < count; k++)
  {
    int16_t y;
    if (d.decimate(in[k], &y))
    {
      out[n++] = y;
    }
  }
  return n;
}

// Interpolator_int16 over a buffer; out must hold count*factor samples.
void ref_interpolate(const float *coefficientsPtr, int taps, int factor,
                     const int16_t *in, uint32_t count, int16_t *out)
{
  std::vector<float> c(coefficientsPtr, coefficientsPtr + taps);
  Interpolator_int16 p(taps, c.data(), factor);
  for (uint32_t k = 0; k < count; k++)
  {
    p.interpolate(in[k], &out[k * factor]);
  }
}

// IirFilter over a buffer (float Direct-Form-I, IirFilter.cc:161).
void ref_iir(const float *b, int nb, const float *a, int na,
             const float *in, uint32_t count, float *out)
{
  std::vector<float> bb(b, b + nb), aa(a, a + na);
  IirFilter f(nb, bb.data(), na, aa.data());
  for (uint32_t k = 0; k < count; k++)
  {
    out[k] = f.filterData(in[k]);
  }
}

// DbfsCalculator's decibel table (DbfsCalculator.cc:58-65), word length 7.
void ref_dbfs_table(int32_t *out)
{
  DbfsCalculator c(7);
  memcpy(out, c.dbTable, sizeof(c.dbTable));
}

int32_t ref_magnitude_to_dbfs(uint32_t magnitude)
{
  DbfsCalculator c(7);
  return c.convertMagnitudeToDbFs(magnitude);
}

// UdpClient::sendData (UdpClient.cc:173-241): the wire format of `enable iqdump`
int ref_udp_send(const char *ip, int port, void *buffer, int length)
{
  UdpClient c(const_cast<char *>(ip), port);
  return c.sendData(buffer, length) ? 1 : 0;
}

// DataProvider (DataProvider.cc:174-231, 235-300): cyclic playback of an .iq file
void *ref_provider_create(void) { return new DataProvider(); }
void ref_provider_destroy(void *p) { delete static_cast<DataProvider *>(p); }
int ref_provider_load(void *p, const char *path) { return static_cast<DataProvider *>(p)->loadIqFile(const_cast<char *>(path)) ? 1 : 0; }
void ref_provider_get(void *p, int8_t *out, uint32_t bytes) { static_cast<DataProvider *>(p)->getIqData(out, bytes); }

// The UB-dependent float -> int16 narrowing, as the host compiler does it.
int16_t ref_float_to_int16(float value)
{
  volatile float v = value;
  return (int16_t)v;
}

} // extern "C"
