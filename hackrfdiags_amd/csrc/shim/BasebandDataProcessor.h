// BasebandDataProcessor.h -- drop-in replacement header: same class name and public interface as
// radioDiags/hdr_diags/BasebandDataProcessor.h:20-43 of the reference (the transmit boundary's dispatcher,
// SURVEY 8a row T5), implemented over the C ABI of libhrfd.so: the PCM ring is hrfd_txring_* (same slots, start
// table and pacing policy as BasebandDataProcessor.cc:416-606), the modulators are the shim's
// Am/Fm/WbFm/SsbModulator classes (hrfd_mod_*).
#ifndef HRFD_SHIM_BASEBANDDATAPROCESSOR_H
#define HRFD_SHIM_BASEBANDDATAPROCESSOR_H

#include <stdint.h>
#include <pthread.h>

#include "hrfd.h"
#include "AmModulator.h"
#include "FmModulator.h"
#include "WbFmModulator.h"
#include "SsbModulator.h"

#define PCM_BLOCK_SIZE (512)
#define PCM_RING_SIZE (16)

class BasebandDataProcessor
{
  public:

  enum modulatorType {None=0, Am=1, Fm=2, WbFm = 3, Lsb = 4, Usb = 5};
  enum streamStateType {Idle, Running};

  BasebandDataProcessor(void);
  ~BasebandDataProcessor(void);

  void setModulatorMode(modulatorType mode);
  void setAmModulator(AmModulator *modulatorPtr);
  void setFmModulator(FmModulator *modulatorPtr);
  void setWbFmModulator(WbFmModulator *modulatorPtr);
  void setSsbModulator(SsbModulator *modulatorPtr);

  void start(void);
  void stop(void);
  void getIqData(int8_t *bufferPtr,int32_t bufferLength);

  void displayInternalInformation(void);

  // not in the reference: the producer side of the ring without the stdin reader thread (tests; an application
  // that has its PCM from somewhere else than standard input)
  void putPcmBlock(const int16_t *pcm512);
  void startWithoutReader(void);        // the stream runs, the ring is filled through putPcmBlock only

  private:

  void modulateBasebandData(int8_t *bufferPtr,uint32_t bufferLength);
  static void *basebandReaderProcedure(void *arg);

  hrfd_txring *ring;
  streamStateType streamState;
  modulatorType modulatorMode;
  AmModulator *amModulatorPtr;
  FmModulator *fmModulatorPtr;
  WbFmModulator *wbFmModulatorPtr;
  SsbModulator *ssbModulatorPtr;
  volatile bool timeToStopReaderThread;
  bool readerThreadStarted;
  pthread_t basebandReaderThread;
  int16_t pcmBlock[PCM_BLOCK_SIZE];
};

#endif
