// BasebandDataProcessor.h -- drop-in replacement header: same class name and public interface as
// radioDiags/hdr_diags/BasebandDataProcessor.h:20-43 of the reference (the transmit boundary's dispatcher,
// SURVEY 8a row T5), implemented over the C ABI of libhrfd.so: the PCM ring is hrfd_txring_* (same slots, start
// table and pacing policy as BasebandDataProcessor.cc:416-606), the modulators are the shim's
// Am/Fm/WbFm/SsbModulator classes (hrfd_mod_*).
// The include guard is the REFERENCE header's own: in a translation unit that has already seen the reference's
// declaration of this class (Radio.h includes its neighbours by quoted name) this header must be a no-op, and the
// other way round; the two declarations are interchangeable by construction (hrfd_shim_layout.h).
#ifndef _BASEBANDDATAPROCESSOR_H_
#define _BASEBANDDATAPROCESSOR_H_
#define HRFD_SHIM_DECLARES_BASEBANDDATAPROCESSOR 1

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

  // BasebandDataProcessor.cc:630-697 (modulateBasebandData): one PCM block from the ring through the mode's modulator
  void fill_transfer_buffer(int8_t *bufferPtr,uint32_t bufferLength);
  static void *reader_main(void *arg);

  hrfd_txring *ring;
  streamStateType running_state;
  modulatorType mode_now;
  AmModulator *am_mod;
  FmModulator *fm_mod;
  WbFmModulator *wbfm_mod;
  SsbModulator *ssb_mod;
  volatile bool reader_must_stop;
  bool reader_running;
  pthread_t reader_thread;
  int16_t one_block[PCM_BLOCK_SIZE];
};

#endif
