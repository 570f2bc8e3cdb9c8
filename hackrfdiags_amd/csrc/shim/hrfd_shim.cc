// hrfd_shim.cc -- the reference-named C++ classes (WbFmDemodulator, FmDemodulator,
// AmDemodulator, SsbDemodulator, IqDataProcessor, SsbModulator, Nco) implemented
// over the C ABI of libhrfd.so.  Plain host C++ (g++), no HIP in here: link with
//     g++ ... hrfd_shim.cc -I include -I hackrfdiags_amd/csrc/shim -lhrfd -lamdhip64
// in place of lib/lib{Am,Fm,WbFm,Ssb}Demodulator.a, libSsbModulator.a and
// src_diags/IqDataProcessor.cc (radioDiags/buildRadioDiags.sh:50-66).
// Like the reference, the classes have no error channel: a failing C-ABI call (no device, out of memory, a HIP
// error) is reported on stderr and aborts (there is no CPU fallback to hide behind).  A buffer LENGTH never aborts:
// acceptIqData takes whatever byteCount the reference's takes (sizeClip / the notes there).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "AmDemodulator.h"
#include "FmDemodulator.h"
#include "WbFmDemodulator.h"
#include "SsbDemodulator.h"
#include "IqDataProcessor.h"
#include "SsbModulator.h"
#include "AmModulator.h"
#include "FmModulator.h"
#include "WbFmModulator.h"
#include "Nco.h"
#include "DataProvider.h"
#include "BasebandDataProcessor.h"

// This file defines the member functions of the SHIM's class declarations.  Its quoted includes find the shim's headers
// beside it before any -I directory; should a build arrange otherwise (the headers share the reference's include
// guards, so the first one seen wins), stop here instead of compiling member functions against the wrong layout.
#if !defined(HRFD_SHIM_DECLARES_IQDATAPROCESSOR) || !defined(HRFD_SHIM_DECLARES_BASEBANDDATAPROCESSOR) || \
    !defined(HRFD_SHIM_DECLARES_DATAPROVIDER) || !defined(HRFD_SHIM_DECLARES_WBFMDEMODULATOR) || \
    !defined(HRFD_SHIM_DECLARES_FMDEMODULATOR) || !defined(HRFD_SHIM_DECLARES_AMDEMODULATOR) || \
    !defined(HRFD_SHIM_DECLARES_SSBDEMODULATOR) || !defined(HRFD_SHIM_DECLARES_AMMODULATOR) || \
    !defined(HRFD_SHIM_DECLARES_FMMODULATOR) || !defined(HRFD_SHIM_DECLARES_WBFMMODULATOR) || \
    !defined(HRFD_SHIM_DECLARES_SSBMODULATOR) || !defined(HRFD_SHIM_DECLARES_NCO)
#error "hrfd_shim.cc must see the shim's own class declarations (hackrfdiags_amd/csrc/shim/*.h), not the reference's"
#endif

// symbols of the host application the reference code also expects
// (Radio.cc:15, diagUi.cc:2881)
extern uint32_t radio_adjustableReceiveGainInDb;
extern void nprintf(FILE *s,const char *formatPtr, ...);

namespace hrfd_shim {

void fatal(const char *what, int rc)
{
  fprintf(stderr, "libhrfd: %s failed (%d): %s\n", what, rc, hrfd_last_error());
  abort();
}

// A buffer longer than the reference's fixed arrays: the reference overruns them (its caller, DataConsumer::acceptData,
// clips first: DataConsumer.cc:229-233); the shim clips and says so once.
uint32_t sizeClip(const char *what, unsigned long byteCount, uint32_t limit)
{
  static bool told = false;
  if (byteCount > limit)
  {
    if (!told)
    {
      told = true;
      nprintf(stderr, "libhrfd: %s: %lu bytes clipped to %u (the reference's arrays hold no more)\n", what, byteCount, limit);
    }
    return limit;
  }
  return (uint32_t)byteCount;
}

DemodulatorBase::DemodulatorBase(int mode, float defaultGain,
                                 void (*pcmCallbackPtr)(int16_t *bufferPtr,uint32_t bufferLength))
{
  this->mode = mode;
  this->gain = defaultGain;
  this->handle = NULL;
  this->pcmCallbackPtr = pcmCallbackPtr;
  memset(pcmData, 0, sizeof(pcmData));
}

DemodulatorBase::~DemodulatorBase(void)
{
  if (handle != NULL)
  {
    hrfd_demod_destroy(handle);
  }
}

void DemodulatorBase::reset(void)
{
  if (handle != NULL)
  {
    int rc = hrfd_demod_reset(handle, 0);
    if (rc != HRFD_OK) fatal("hrfd_demod_reset", rc);
  }
}

void DemodulatorBase::setGain(float gain)
{
  this->gain = gain;
  if (handle != NULL)
  {
    int rc = hrfd_demod_set_gain(handle, 0, gain);
    if (rc != HRFD_OK) fatal("hrfd_demod_set_gain", rc);
  }
}

void DemodulatorBase::setSideband(bool lsb)
{
  mode = lsb ? HRFD_MODE_LSB : HRFD_MODE_USB;
  if (handle != NULL)
  {
    int rc = hrfd_demod_set_sideband(handle, 0, lsb ? 1 : 0);
    if (rc != HRFD_OK) fatal("hrfd_demod_set_sideband", rc);
  }
}

void DemodulatorBase::accept(int8_t *bufferPtr,uint32_t bufferLength)
{
  uint32_t sampleCount = 0;
  int rc;

  if (handle == NULL)
  {
    rc = hrfd_demod_create(mode, 1, -1, &handle);
    if (rc != HRFD_OK) fatal("hrfd_demod_create", rc);
    rc = hrfd_demod_set_gain(handle, 0, gain);
    if (rc != HRFD_OK) fatal("hrfd_demod_set_gain", rc);
  }
  // Lengths, as the reference's loops treat them.  More than the member arrays hold (32768 bytes) overruns them
  // there: clipped here.  An odd count: WbFmDemodulator::demodulateSignal takes bufferLength / 2 samples
  // (WbFmDemodulator.cc:395); the other three run their Q loop to bufferPtr[bufferLength] (FmDemodulator.cc:426,
  // AmDemodulator.cc:381, SsbDemodulator.cc:502), i.e. (bufferLength + 1) / 2 samples.
  bufferLength = sizeClip("acceptIqData", bufferLength, 32768u);
  if ((bufferLength & 1u) != 0)
  {
    bufferLength = (mode == HRFD_MODE_WBFM) ? bufferLength - 1u : bufferLength + 1u;
  }
  if (bufferLength != 0)
  {
    // 256 kS/s IQ in, PCM out; the reference's arrays hold at most 32768 bytes / 512 samples
    rc = hrfd_demod_process(handle, bufferPtr, bufferLength, pcmData, &sampleCount);
    if (rc != HRFD_OK) fatal("hrfd_demod_process", rc);
  }
  // sendPcmData: synchronously, on the caller's thread, buffer valid during the call -- with whatever count the
  // decimators completed, 0 included (WbFmDemodulator.cc:341-356, :520-529)
  pcmCallbackPtr(pcmData, sampleCount);
}

void DemodulatorBase::deliverPcm(int16_t *bufferPtr,uint32_t bufferLength)
{
  pcmCallbackPtr(bufferPtr, bufferLength);
}

void DemodulatorBase::display(const char *name)
{
  nprintf(stderr, "\n--------------------------------------------\n");
  nprintf(stderr, "%s Internal Information (libhrfd, MI355X)\n", name);
  nprintf(stderr, "--------------------------------------------\n");
  nprintf(stderr, "Demodulator Gain         : %f\n", gain);
}

} // namespace hrfd_shim

using hrfd_shim::fatal;

// ---------------------------------------------------------------- demodulators
// default gains: WbFmDemodulator.cc:151, FmDemodulator.cc:173, AmDemodulator.cc:102
WbFmDemodulator::WbFmDemodulator(void (*pcmCallbackPtr)(int16_t *bufferPtr,uint32_t bufferLength))
  : DemodulatorBase(HRFD_MODE_WBFM, (float)(256000/(2 * M_PI)), pcmCallbackPtr) {}
WbFmDemodulator::~WbFmDemodulator(void) {}
void WbFmDemodulator::resetDemodulator(void) { reset(); }
void WbFmDemodulator::setDemodulatorGain(float gain) { setGain(gain); }
void WbFmDemodulator::acceptIqData(int8_t *bufferPtr,uint32_t bufferLength) { accept(bufferPtr, bufferLength); }
void WbFmDemodulator::displayInternalInformation(void) { display("Wideband FM Demodulator"); }

FmDemodulator::FmDemodulator(void (*pcmCallbackPtr)(int16_t *bufferPtr,uint32_t bufferLength))
  : DemodulatorBase(HRFD_MODE_FM, (float)(64000/(2 * M_PI)), pcmCallbackPtr) {}
FmDemodulator::~FmDemodulator(void) {}
void FmDemodulator::resetDemodulator(void) { reset(); }
void FmDemodulator::setDemodulatorGain(float gain) { setGain(gain); }
void FmDemodulator::acceptIqData(int8_t *bufferPtr,uint32_t bufferLength) { accept(bufferPtr, bufferLength); }
void FmDemodulator::displayInternalInformation(void) { display("FM Demodulator"); }

AmDemodulator::AmDemodulator(void (*pcmCallbackPtr)(int16_t *bufferPtr,uint32_t bufferLength))
  : DemodulatorBase(HRFD_MODE_AM, 300, pcmCallbackPtr) {}
AmDemodulator::~AmDemodulator(void) {}
void AmDemodulator::resetDemodulator(void) { reset(); }
void AmDemodulator::setDemodulatorGain(float gain) { setGain(gain); }
void AmDemodulator::acceptIqData(int8_t *bufferPtr,uint32_t bufferLength) { accept(bufferPtr, bufferLength); }
void AmDemodulator::displayInternalInformation(void) { display("AM Demodulator"); }

SsbDemodulator::SsbDemodulator(void (*pcmCallbackPtr)(int16_t *bufferPtr,uint32_t bufferLength))
  : DemodulatorBase(HRFD_MODE_LSB, 300, pcmCallbackPtr) { lsb = true; }
SsbDemodulator::~SsbDemodulator(void) {}
void SsbDemodulator::resetDemodulator(void) { reset(); }
void SsbDemodulator::setLsbDemodulationMode(void) { lsb = true; setSideband(true); }
void SsbDemodulator::setUsbDemodulationMode(void) { lsb = false; setSideband(false); }
void SsbDemodulator::setDemodulatorGain(float gain) { setGain(gain); }
void SsbDemodulator::acceptIqData(int8_t *bufferPtr,uint32_t bufferLength) { accept(bufferPtr, bufferLength); }
void SsbDemodulator::displayInternalInformation(void) { display("SSB Demodulator"); }

// ---------------------------------------------------------------- IqDataProcessor
IqDataProcessor::IqDataProcessor(char *hostIpAddress,int hostPort)
{
  // IqDataProcessor.cc:139: the `enable iqdump` stream goes to hostIpAddress:hostPort as
  // 2048-byte datagrams (the application's UdpClient) unless a sink is registered instead
  networkInterfacePtr = new UdpClient(hostIpAddress, hostPort);
  handle = NULL;
  demodulatorMode = None;
  signalDetectThreshold = -200;
  amDemodulatorPtr = NULL;
  fmDemodulatorPtr = NULL;
  wbFmDemodulatorPtr = NULL;
  ssbDemodulatorPtr = NULL;
  iqDumpEnabled = false;
  iqDumpSinkPtr = NULL;
  iqDumpContextPtr = NULL;
  signalNotificationEnabled = false;
  signalCallbackPtr = NULL;
  signalCallbackContextPtr = NULL;
  signalMagnitudeNotificationEnabled = false;
  signalMagnitudeCallbackPtr = NULL;
  signalMagnitudeCallbackContextPtr = NULL;
  work = new Work();
  memset(work, 0, sizeof(Work));
  for (int i = 0; i < 4; i++) work->pushedGain[i] = nanf("");
}

IqDataProcessor::~IqDataProcessor(void)
{
  if (networkInterfacePtr != NULL)
  {
    delete networkInterfacePtr;
  }
  if (handle != NULL)
  {
    hrfd_rx_destroy(handle);
  }
  delete work;
}

void IqDataProcessor::ensureHandle(void)
{
  if (handle == NULL)
  {
    int rc = hrfd_rx_create(1, -1, &handle);
    if (rc != HRFD_OK) fatal("hrfd_rx_create", rc);
    rc = hrfd_rx_set_mode(handle, 0, (int)demodulatorMode);
    if (rc != HRFD_OK) fatal("hrfd_rx_set_mode", rc);
    rc = hrfd_rx_set_threshold(handle, 0, signalDetectThreshold);
    if (rc != HRFD_OK) fatal("hrfd_rx_set_threshold", rc);
  }
}

// the demodulator objects own the gain setting in the reference; mirror it into
// the fused chain whenever it changed (the CLI thread calls setDemodulatorGain on them)
void IqDataProcessor::pushGains(void)
{
  const hrfd_shim::DemodulatorBase *d[4] = {amDemodulatorPtr, fmDemodulatorPtr, wbFmDemodulatorPtr,
                                            ssbDemodulatorPtr};
  const int modes[4] = {HRFD_MODE_AM, HRFD_MODE_FM, HRFD_MODE_WBFM, HRFD_MODE_LSB};
  for (int i = 0; i < 4; i++)
  {
    if (d[i] != NULL && !(d[i]->currentGain() == work->pushedGain[i]))
    {
      work->pushedGain[i] = d[i]->currentGain();
      int rc = hrfd_rx_set_gain(handle, 0, modes[i], work->pushedGain[i]);
      if (rc != HRFD_OK) fatal("hrfd_rx_set_gain", rc);
    }
  }
}

void IqDataProcessor::setDemodulatorMode(demodulatorType mode)
{
  demodulatorMode = mode;
  if (mode == Lsb && ssbDemodulatorPtr != NULL) ssbDemodulatorPtr->setLsbDemodulationMode();
  if (mode == Usb && ssbDemodulatorPtr != NULL) ssbDemodulatorPtr->setUsbDemodulationMode();
  if (handle != NULL)
  {
    int rc = hrfd_rx_set_mode(handle, 0, (int)mode);
    if (rc != HRFD_OK) fatal("hrfd_rx_set_mode", rc);
  }
}

void IqDataProcessor::setAmDemodulator(AmDemodulator *demodulatorPtr) { amDemodulatorPtr = demodulatorPtr; }
void IqDataProcessor::setFmDemodulator(FmDemodulator *demodulatorPtr) { fmDemodulatorPtr = demodulatorPtr; }
void IqDataProcessor::setWbFmDemodulator(WbFmDemodulator *demodulatorPtr) { wbFmDemodulatorPtr = demodulatorPtr; }
void IqDataProcessor::setSsbDemodulator(SsbDemodulator *demodulatorPtr) { ssbDemodulatorPtr = demodulatorPtr; }

void IqDataProcessor::setSignalDetectThreshold(int32_t threshold)
{
  signalDetectThreshold = threshold;
  if (handle != NULL)
  {
    int rc = hrfd_rx_set_threshold(handle, 0, threshold);
    if (rc != HRFD_OK) fatal("hrfd_rx_set_threshold", rc);
  }
}

void IqDataProcessor::acceptIqData(unsigned long timeStamp,
                                   int8_t *bufferPtr,
                                   unsigned long byteCount)
{
  uint32_t sampleCount = 0, signalMagnitude = 0;
  uint8_t signalAllowed = 0;
  int16_t *const pcmData = work->pcmData;
  int8_t *const decimatedData = work->decimatedData;
  (void)timeStamp;

  ensureHandle();
  pushGains();
  // Lengths: whatever the reference takes.  DataConsumer passes short blocks on (DataConsumer.cc:229-241) and the
  // decimators keep their commutator positions between calls, so any count works there and here; longer than the
  // arrays is clipped (as DataConsumer does before it calls); an odd count makes the reference's Q loop read
  // bufferPtr[byteCount] (IqDataProcessor.cc:474): the same byte is read here.  Nothing at all: the reference divides
  // by zero in SignalDetector.cc:255; here the call is ignored.
  uint32_t length = hrfd_shim::sizeClip("IqDataProcessor::acceptIqData", byteCount, HRFD_BLOCK_BYTES);
  length += (length & 1u);
  if (length == 0)
  {
    return;
  }
  uint32_t pending = 0;
  int rc = hrfd_rx_pending_samples(handle, &pending);
  if (rc != HRFD_OK) fatal("hrfd_rx_pending_samples", rc);
  const uint32_t decimatedByteCount = 2u * ((pending + length / 2u) / 8u);     // reduceSampleRate's return value
  // reduceSampleRate + upconvertByFsOver4 + Squelch::run + demodulator, one launch
  rc = hrfd_rx_process_block(handle, bufferPtr, length, 1,
                             radio_adjustableReceiveGainInDb, pcmData, &sampleCount,
                             &signalMagnitude, &signalAllowed,
                             iqDumpEnabled ? decimatedData : NULL);
  if (rc != HRFD_OK) fatal("hrfd_rx_process_block", rc);

  // same order as the reference (IqDataProcessor.cc:953-1034)
  if (iqDumpEnabled)
  {
    if (iqDumpSinkPtr != NULL)
    {
      iqDumpSinkPtr(decimatedData, decimatedByteCount, iqDumpContextPtr);
    }
    else
    {
      networkInterfacePtr->sendData(decimatedData, (int)decimatedByteCount);   // IqDataProcessor.cc:956
    }
  }
  if (signalNotificationEnabled && signalCallbackPtr != NULL)
  {
    signalCallbackPtr(signalAllowed != 0, signalCallbackContextPtr);
  }
  if (signalMagnitudeNotificationEnabled && signalMagnitudeCallbackPtr != NULL)
  {
    signalMagnitudeCallbackPtr(signalMagnitude, signalMagnitudeCallbackContextPtr);
  }
  if (signalAllowed)
  {
    hrfd_shim::DemodulatorBase *d = NULL;
    switch (demodulatorMode)
    {
      case Am: d = amDemodulatorPtr; break;
      case Fm: d = fmDemodulatorPtr; break;
      case WbFm: d = wbFmDemodulatorPtr; break;
      case Lsb:
      case Usb: d = ssbDemodulatorPtr; break;
      default: break;
    }
    if (d != NULL)
    {
      // X::acceptIqData ends in sendPcmData with whatever the decimators completed -- a short block may complete
      // nothing and the reference still calls back, with a count of 0 (e.g. WbFmDemodulator.cc:341-356)
      d->deliverPcm(pcmData, sampleCount);
    }
  }
}

// Public in the reference (IqDataProcessor.cc:429-500): the three half-band stages per rail over one buffer,
// result in the private decimatedData, decimator pipelines advanced, NO Fs/4 mix (acceptIqData applies that
// separately), squelch and demodulators untouched.  hrfd_rx_reduce_sample_rate does that on the device, where the
// front end only exists fused with the mixer: the mix is taken out of decimatedData again -- the int8 rotation is
// exactly invertible, -(-128) wraps to -128 both ways.
uint32_t IqDataProcessor::reduceSampleRate(int8_t *bufferPtr,uint32_t bufferLength)
{
  ensureHandle();
  uint32_t length = hrfd_shim::sizeClip("IqDataProcessor::reduceSampleRate", bufferLength, HRFD_BLOCK_BYTES);
  length += (length & 1u);                                   // the Q loop reads bufferPtr[bufferLength] (:474)
  if (length == 0)
  {
    return 0;
  }
  uint32_t pending = 0;
  int rc = hrfd_rx_pending_samples(handle, &pending);
  if (rc != HRFD_OK) fatal("hrfd_rx_pending_samples", rc);
  const uint32_t byteCount = 2u * ((pending + length / 2u) / 8u);
  rc = hrfd_rx_reduce_sample_rate(handle, bufferPtr, length, work->decimatedData);
  if (rc != HRFD_OK) fatal("reduceSampleRate", rc);
  downconvertByFsOver4(work->decimatedData, byteCount);
  return byteCount;
}

// Stand-alone helpers of the reference's public interface (IqDataProcessor.h:55-56): multiply sample n of a
// caller-owned interleaved int8 IQ buffer by j^n (up) or (-j)^n (down).  Not on the per-block path (the fused
// kernels rotate by themselves).  One table-free form for both directions: a quarter turn maps (i, q) to (-q, i),
// k quarter turns are applied by exchanging and negating according to k & 3; int8 negation wraps.
namespace {
inline void quarter_turns(int8_t *iq, uint32_t byteCount, unsigned step)
{
  const uint32_t samples = byteCount / 2;
  for (uint32_t n = 0; n < samples; n++)
  {
    const unsigned k = (n * step) & 3u;                  // quarter turns for this sample
    const int8_t i = iq[2 * n], q = iq[2 * n + 1];
    const int8_t a = (k & 1u) ? q : i, b = (k & 1u) ? i : q;            // odd k exchanges the rails
    const bool negFirst = (k == 1u || k == 2u), negSecond = (k == 2u || k == 3u);
    iq[2 * n] = negFirst ? (int8_t)(0 - a) : a;
    iq[2 * n + 1] = negSecond ? (int8_t)(0 - b) : b;
  }
}
}  // namespace

void IqDataProcessor::upconvertByFsOver4(int8_t *bufferPtr,uint32_t byteCount)
{
  quarter_turns(bufferPtr, byteCount, 1u);               // j^n
}

void IqDataProcessor::downconvertByFsOver4(int8_t *bufferPtr,uint32_t byteCount)
{
  quarter_turns(bufferPtr, byteCount, 3u);               // (-j)^n = j^(3n)
}

void IqDataProcessor::enableSignalNotification(void) { signalNotificationEnabled = true; }
void IqDataProcessor::disableSignalNotification(void) { signalNotificationEnabled = false; }
void IqDataProcessor::registerSignalStateCallback(
    void (*signalCallbackPtr)(bool signalPresent,void *contextPtr), void *contextPtr)
{
  this->signalCallbackContextPtr = contextPtr;
  this->signalCallbackPtr = signalCallbackPtr;
}
void IqDataProcessor::enableSignalMagnitudeNotification(void) { signalMagnitudeNotificationEnabled = true; }
void IqDataProcessor::disableSignalMagnitudeNotification(void) { signalMagnitudeNotificationEnabled = false; }
void IqDataProcessor::registerSignalMagnitudeCallback(
    void (*callbackPtr)(uint32_t signalMagnitude,void *contextPtr), void *contextPtr)
{
  this->signalMagnitudeCallbackContextPtr = contextPtr;
  this->signalMagnitudeCallbackPtr = callbackPtr;
}
void IqDataProcessor::enableIqDump(void) { iqDumpEnabled = true; }
void IqDataProcessor::disableIqDump(void) { iqDumpEnabled = false; }
bool IqDataProcessor::isIqDumpEnabled(void) { return iqDumpEnabled; }
void IqDataProcessor::registerIqDumpSink(
    void (*sinkPtr)(int8_t *bufferPtr,uint32_t byteCount,void *contextPtr), void *contextPtr)
{
  iqDumpSinkPtr = sinkPtr;
  iqDumpContextPtr = contextPtr;
}

void IqDataProcessor::displayInternalInformation(void)
{
  static const char *names[] = {"None", "AM", "FM", "WBFM", "LSB", "USB"};
  nprintf(stderr, "\n--------------------------------------------\n");
  nprintf(stderr, "IqDataProcessor Internal Information (libhrfd, MI355X)\n");
  nprintf(stderr, "--------------------------------------------\n");
  nprintf(stderr, "Demodulator Mode         : %s\n", names[(int)demodulatorMode]);
  nprintf(stderr, "Signal Detect Threshold  : %d dBFs\n", signalDetectThreshold);
  nprintf(stderr, "IQ Dump                  : %s\n", iqDumpEnabled ? "Enabled" : "Disabled");
}

// ---------------------------------------------------------------- SsbModulator
SsbModulator::SsbModulator(void)
{
  lsbModulationMode = true;
  handle = NULL;
}

SsbModulator::~SsbModulator(void)
{
  if (handle != NULL)
  {
    hrfd_mod_destroy(handle);
  }
}

void SsbModulator::resetModulator(void)
{
  if (handle != NULL)
  {
    int rc = hrfd_mod_reset(handle, 0);
    if (rc != HRFD_OK) fatal("hrfd_mod_reset", rc);
  }
}

void SsbModulator::setLsbModulationMode(void)
{
  lsbModulationMode = true;
  if (handle != NULL) hrfd_mod_set_sideband(handle, 0, 1);
}

void SsbModulator::setUsbModulationMode(void)
{
  lsbModulationMode = false;
  if (handle != NULL) hrfd_mod_set_sideband(handle, 0, 0);
}

void SsbModulator::acceptData(int16_t *bufferPtr,
                              uint32_t bufferLength,
                              int8_t *outputBufferPtr,
                              uint32_t *outputBufferLengthPtr)
{
  int rc;
  if (handle == NULL)
  {
    rc = hrfd_mod_create(HRFD_MOD_SSB, 1, -1, &handle);
    if (rc != HRFD_OK) fatal("hrfd_mod_create", rc);
    hrfd_mod_set_sideband(handle, 0, lsbModulationMode ? 1 : 0);
  }
  rc = hrfd_mod_process(handle, bufferPtr, bufferLength, outputBufferPtr, outputBufferLengthPtr);
  if (rc != HRFD_OK) fatal("hrfd_mod_process", rc);
}

void SsbModulator::displayInternalInformation(void)
{
  nprintf(stderr, "\n--------------------------------------------\n");
  nprintf(stderr, "SSB Modulator Internal Information (libhrfd, MI355X)\n");
  nprintf(stderr, "--------------------------------------------\n");
  nprintf(stderr, "Modulation Mode          : %s\n", lsbModulationMode ? "LSB" : "USB");
}

// ---------------------------------------------------------------- AmModulator
AmModulator::AmModulator(void)
{
  modulationIndex = 0.8;                                 // AmModulator.cc:218
  handle = NULL;
}

AmModulator::~AmModulator(void)
{
  if (handle != NULL)
  {
    hrfd_mod_destroy(handle);
  }
}

void AmModulator::resetModulator(void)
{
  if (handle != NULL)
  {
    int rc = hrfd_mod_reset(handle, 0);
    if (rc != HRFD_OK) fatal("hrfd_mod_reset", rc);
  }
}

void AmModulator::setModulationIndex(float modulationIndex)
{
  if ((modulationIndex >= 0) && (modulationIndex <= 1))  // AmModulator.cc:332
  {
    this->modulationIndex = modulationIndex;
  }
  if (handle != NULL) hrfd_mod_set_modulation_index(handle, 0, this->modulationIndex);
}

void AmModulator::acceptData(int16_t *bufferPtr,
                             uint32_t bufferLength,
                             int8_t *outputBufferPtr,
                             uint32_t *outputBufferLengthPtr)
{
  int rc;
  if (handle == NULL)
  {
    rc = hrfd_mod_create(HRFD_MOD_AM, 1, -1, &handle);
    if (rc != HRFD_OK) fatal("hrfd_mod_create", rc);
    hrfd_mod_set_modulation_index(handle, 0, modulationIndex);
  }
  rc = hrfd_mod_process(handle, bufferPtr, bufferLength, outputBufferPtr, outputBufferLengthPtr);
  if (rc != HRFD_OK) fatal("hrfd_mod_process", rc);
}

void AmModulator::displayInternalInformation(void)
{
  nprintf(stderr, "\n--------------------------------------------\n");
  nprintf(stderr, "AM Modulator Internal Information (libhrfd, MI355X)\n");
  nprintf(stderr, "--------------------------------------------\n");
  nprintf(stderr, "Modulator Index          : %f\n", modulationIndex);
}

// ---------------------------------------------------------------- FmModulator
FmModulator::FmModulator(void)
{
  frequencyDeviation = 3500;                             // FmModulator.cc:218
  handle = NULL;
}

FmModulator::~FmModulator(void)
{
  if (handle != NULL)
  {
    hrfd_mod_destroy(handle);
  }
}

void FmModulator::resetModulator(void)
{
  if (handle != NULL)
  {
    int rc = hrfd_mod_reset(handle, 0);
    if (rc != HRFD_OK) fatal("hrfd_mod_reset", rc);
  }
}

void FmModulator::setFrequencyDeviation(float deviaton)
{
  // FmModulator.cc:339 tests the member, not the argument: kept
  if ((frequencyDeviation >= 0) && (frequencyDeviation <= 3500))
  {
    this->frequencyDeviation = deviaton;
  }
  if (handle != NULL) hrfd_mod_set_deviation(handle, 0, deviaton);
}

void FmModulator::acceptData(int16_t *bufferPtr,
                             uint32_t bufferLength,
                             int8_t *outputBufferPtr,
                             uint32_t *outputBufferLengthPtr)
{
  int rc;
  if (handle == NULL)
  {
    rc = hrfd_mod_create(HRFD_MOD_FM, 1, -1, &handle);
    if (rc != HRFD_OK) fatal("hrfd_mod_create", rc);
    // replay the setter history into the fresh handle: one call reproduces any reachable value
    if (frequencyDeviation != 3500) hrfd_mod_set_deviation(handle, 0, frequencyDeviation);
  }
  rc = hrfd_mod_process(handle, bufferPtr, bufferLength, outputBufferPtr, outputBufferLengthPtr);
  if (rc != HRFD_OK) fatal("hrfd_mod_process", rc);
}

void FmModulator::displayInternalInformation(void)
{
  nprintf(stderr, "\n--------------------------------------------\n");
  nprintf(stderr, "FM Modulator Internal Information (libhrfd, MI355X)\n");
  nprintf(stderr, "--------------------------------------------\n");
  nprintf(stderr, "Frequency Deviation:      : %fHz\n", frequencyDeviation);
}

// ---------------------------------------------------------------- WbFmModulator (wideband)
WbFmModulator::WbFmModulator(void)
{
  frequencyDeviation = 70000;                            // WbFmModulator.cc:204
  handle = NULL;
}

WbFmModulator::~WbFmModulator(void)
{
  if (handle != NULL)
  {
    hrfd_mod_destroy(handle);
  }
}

void WbFmModulator::resetModulator(void)
{
  if (handle != NULL)
  {
    int rc = hrfd_mod_reset(handle, 0);
    if (rc != HRFD_OK) fatal("hrfd_mod_reset", rc);
  }
}

void WbFmModulator::setFrequencyDeviation(float deviaton)
{
  // WbFmModulator.cc:313 tests the member, not the argument: kept
  if ((frequencyDeviation >= 0) && (frequencyDeviation <= 112000))
  {
    this->frequencyDeviation = deviaton;
  }
  if (handle != NULL) hrfd_mod_set_deviation(handle, 0, deviaton);
}

void WbFmModulator::acceptData(int16_t *bufferPtr,
                             uint32_t bufferLength,
                             int8_t *outputBufferPtr,
                             uint32_t *outputBufferLengthPtr)
{
  int rc;
  if (handle == NULL)
  {
    rc = hrfd_mod_create(HRFD_MOD_WBFM, 1, -1, &handle);
    if (rc != HRFD_OK) fatal("hrfd_mod_create", rc);
    // replay the setter history into the fresh handle: one call reproduces any reachable value
    if (frequencyDeviation != 70000) hrfd_mod_set_deviation(handle, 0, frequencyDeviation);
  }
  rc = hrfd_mod_process(handle, bufferPtr, bufferLength, outputBufferPtr, outputBufferLengthPtr);
  if (rc != HRFD_OK) fatal("hrfd_mod_process", rc);
}

void WbFmModulator::displayInternalInformation(void)
{
  nprintf(stderr, "\n--------------------------------------------\n");
  nprintf(stderr, "WBFM Modulator Internal Information (libhrfd, MI355X)\n");
  nprintf(stderr, "--------------------------------------------\n");
  nprintf(stderr, "Frequency Deviation:      : %fHz\n", frequencyDeviation);
}

// ---------------------------------------------------------------- Nco
Nco::Nco(float sampleRate,float frequency)
{
  this->sampleRate = sampleRate;
  this->frequency = frequency;
  handle = NULL;
  int rc = hrfd_nco_create(1, sampleRate, frequency, -1, &handle);
  if (rc != HRFD_OK) fatal("hrfd_nco_create", rc);
}

Nco::~Nco(void)
{
  if (handle != NULL)
  {
    hrfd_nco_destroy(handle);
  }
}

void Nco::setFrequency(float frequency)
{
  this->frequency = frequency;
  int rc = hrfd_nco_set_frequency(handle, 0, frequency);
  if (rc != HRFD_OK) fatal("hrfd_nco_set_frequency", rc);
}

void Nco::reset(void)
{
  int rc = hrfd_nco_reset(handle, 0);
  if (rc != HRFD_OK) fatal("hrfd_nco_reset", rc);
}

void Nco::run(float *iValuePtr,float *qValuePtr)
{
  int rc = hrfd_nco_run(handle, 0, 1, iValuePtr, qValuePtr);
  if (rc != HRFD_OK) fatal("hrfd_nco_run", rc);
}

void Nco::runFast(float *iValuePtr,float *qValuePtr)
{
  int rc = hrfd_nco_run(handle, 1, 1, iValuePtr, qValuePtr);
  if (rc != HRFD_OK) fatal("hrfd_nco_run", rc);
}

// ---------------------------------------------------------------- DataProvider
DataProvider::DataProvider(void)
{
  handle = NULL;
  iqFileName[0] = 0;
}

DataProvider::~DataProvider(void)
{
  if (handle != NULL)
  {
    hrfd_play_destroy(handle);
  }
}

// DataProvider.cc:235-300
bool DataProvider::loadIqFile(char *fileNamePtr)
{
  if (handle == NULL)
  {
    const int rc = hrfd_play_create(1, -1, &handle);
    if (rc != HRFD_OK) hrfd_shim::fatal("hrfd_play_create", rc);
  }
  if (hrfd_play_load_file(handle, fileNamePtr) != HRFD_OK)
  {
    return false;
  }
  strncpy(iqFileName, fileNamePtr, sizeof(iqFileName) - 1);
  iqFileName[sizeof(iqFileName) - 1] = 0;
  return true;
}

// DataProvider.cc:122-131: nothing happens while no file is loaded
void DataProvider::getIqData(int8_t *bufferPtr,uint32_t bufferLength)
{
  if (handle != NULL)
  {
    const int rc = hrfd_play_get(handle, bufferPtr, bufferLength);
    if (rc != HRFD_OK) hrfd_shim::fatal("hrfd_play_get", rc);
  }
}

void DataProvider::displayInternalInformation(void)
{
  uint32_t index = 0;
  if (handle != NULL) (void)hrfd_play_get_position(handle, 0, &index);
  nprintf(stderr, "\n--------------------------------------------\n");
  nprintf(stderr, "Data Provider Internal Information (libhrfd, MI355X)\n");
  nprintf(stderr, "--------------------------------------------\n");
  nprintf(stderr, "IQ File Name            : %s\n", iqFileName);
  nprintf(stderr, "IQ Sample Buffer Index  : %u\n", index);
}

// ---------------------------------------------------------------------------------------------
// BasebandDataProcessor: the transmit boundary's dispatcher.  libhackrf's transmit callback asks
// for one transfer buffer of IQ (getIqData, BasebandDataProcessor.cc:381); one 512-sample PCM block comes off the
// ring (hrfd_txring_read_batch = getNextFilledBuffer, :482-606: drop / repeat pacing, zeros while idle) and goes
// through the modulator of the current mode (:630-697); mode None fills the buffer with 64 (:641).
// ---------------------------------------------------------------------------------------------
#include <sys/select.h>

BasebandDataProcessor::BasebandDataProcessor(void)
    : ring(NULL), running_state(Idle), mode_now(None), am_mod(NULL), fm_mod(NULL),
      wbfm_mod(NULL), ssb_mod(NULL), reader_must_stop(false), reader_running(false),
      reader_thread()
{
  const int rc = hrfd_txring_create(1, &ring);
  if (rc != HRFD_OK) fatal("hrfd_txring_create", rc);
  memset(one_block, 0, sizeof(one_block));
}

BasebandDataProcessor::~BasebandDataProcessor(void)
{
  stop();
  hrfd_txring_destroy(ring);
}

void BasebandDataProcessor::setAmModulator(AmModulator *modulatorPtr) { am_mod = modulatorPtr; }
void BasebandDataProcessor::setFmModulator(FmModulator *modulatorPtr) { fm_mod = modulatorPtr; }
void BasebandDataProcessor::setWbFmModulator(WbFmModulator *modulatorPtr) { wbfm_mod = modulatorPtr; }
void BasebandDataProcessor::setSsbModulator(SsbModulator *modulatorPtr) { ssb_mod = modulatorPtr; }

void BasebandDataProcessor::setModulatorMode(modulatorType mode)
{
  mode_now = mode;
  if (ssb_mod != NULL)
  {
    if (mode == Lsb) ssb_mod->setLsbModulationMode();
    if (mode == Usb) ssb_mod->setUsbModulationMode();
  }
}

void BasebandDataProcessor::putPcmBlock(const int16_t *pcm512)
{
  const int rc = hrfd_txring_write(ring, 0, pcm512);
  if (rc != HRFD_OK) fatal("hrfd_txring_write", rc);
}

// BasebandDataProcessor.cc:834-887: standard input, 512 samples at a time, polled every 5 ms
void *BasebandDataProcessor::reader_main(void *arg)
{
  BasebandDataProcessor *me = static_cast<BasebandDataProcessor *>(arg);
  int16_t block[PCM_BLOCK_SIZE];
  fprintf(stderr, "Entering Baseband Reader.\n");
  while (!me->reader_must_stop)
  {
    fd_set fds;
    FD_ZERO(&fds);
    FD_SET(0, &fds);
    struct timeval tv = {0, 5000};
    if (select(1, &fds, NULL, NULL, &tv) > 0)
    {
      memset(block, 0, sizeof(block));
      if (fread(block, sizeof(int16_t), PCM_BLOCK_SIZE, stdin) == 0 && feof(stdin))
      {
        break;                                           // (the reference keeps polling a closed stdin)
      }
      me->putPcmBlock(block);
    }
  }
  fprintf(stderr, "Exiting Baseband Reader.\n");
  return NULL;
}

void BasebandDataProcessor::start(void)
{
  if (running_state == Idle)
  {
    reader_must_stop = false;
    hrfd_txring_set_running(ring, 0, 1);
    pthread_create(&reader_thread, NULL, reader_main, this);
    reader_running = true;
    running_state = Running;
  }
}

void BasebandDataProcessor::startWithoutReader(void)
{
  if (running_state == Idle)
  {
    hrfd_txring_set_running(ring, 0, 1);
    reader_running = false;
    running_state = Running;
  }
}

void BasebandDataProcessor::stop(void)
{
  if (running_state == Running)
  {
    reader_must_stop = true;
    if (reader_running)
    {
      pthread_join(reader_thread, NULL);
      reader_running = false;
    }
    hrfd_txring_set_running(ring, 0, 0);                 // also drops the ring's synchronisation (:352)
    running_state = Idle;
  }
}

void BasebandDataProcessor::getIqData(int8_t *bufferPtr,int32_t byteCount)
{
  fill_transfer_buffer(bufferPtr, (uint32_t)byteCount);
}

void BasebandDataProcessor::fill_transfer_buffer(int8_t *bufferPtr,uint32_t bufferLength)
{
  uint32_t outputBufferLength = 0;
  const int rc = hrfd_txring_read_batch(ring, one_block);
  if (rc != HRFD_OK) fatal("hrfd_txring_read_batch", rc);
  switch (mode_now)
  {
    case None: memset(bufferPtr, 64, bufferLength); break;
    case Am: am_mod->acceptData(one_block, PCM_BLOCK_SIZE, bufferPtr, &outputBufferLength); break;
    case Fm: fm_mod->acceptData(one_block, PCM_BLOCK_SIZE, bufferPtr, &outputBufferLength); break;
    case WbFm: wbfm_mod->acceptData(one_block, PCM_BLOCK_SIZE, bufferPtr, &outputBufferLength); break;
    case Lsb:
    case Usb: ssb_mod->acceptData(one_block, PCM_BLOCK_SIZE, bufferPtr, &outputBufferLength); break;
    default: break;
  }
}

void BasebandDataProcessor::displayInternalInformation(void)
{
  uint32_t st[6] = {0, 0, 0, 0, 0, 0};
  hrfd_txring_stats(ring, 0, st);
  nprintf(stderr, "\n--------------------------------------------\n");
  nprintf(stderr, "Baseband Data Processor Internal Information\n");
  nprintf(stderr, "--------------------------------------------\n");
  static const char *names[] = {"None", "AM", "FM", "WBFM", "LSB", "USB"};
  nprintf(stderr, "Modulator Mode            : %s\n", names[(int)mode_now]);
  nprintf(stderr, "Stream State              : %s\n", running_state == Running ? "Running" : "Idle");
  nprintf(stderr, "PCM Buffers Produced      : %u\n", st[0]);
  nprintf(stderr, "PCM Buffers Consumed      : %u\n", st[1]);
  nprintf(stderr, "PCM Blocks Dropped        : %u\n", st[2]);
  nprintf(stderr, "PCM Blocks Added          : %u\n", st[3]);
  nprintf(stderr, "PCM Writer Index          : %u\n", st[4]);
  nprintf(stderr, "PCM Reader Index          : %u\n", st[5]);
}

// ---------------------------------------------------------------------------------------------
// Layout containment (hrfd_shim_layout.h): application objects compiled against the reference's own headers may
// allocate these classes; what they allocate must be enough for the shim.
// ---------------------------------------------------------------------------------------------
#include "hrfd_shim_layout.h"

HRFD_SHIM_FITS(IqDataProcessor);
HRFD_SHIM_FITS(BasebandDataProcessor);
HRFD_SHIM_FITS(DataProvider);
HRFD_SHIM_FITS(AmDemodulator);
HRFD_SHIM_FITS(FmDemodulator);
HRFD_SHIM_FITS(WbFmDemodulator);
HRFD_SHIM_FITS(SsbDemodulator);
HRFD_SHIM_FITS(AmModulator);
HRFD_SHIM_FITS(FmModulator);
HRFD_SHIM_FITS(WbFmModulator);
HRFD_SHIM_FITS(SsbModulator);
HRFD_SHIM_FITS(Nco);
