// WbFmDemodulator.h -- drop-in replacement header: same class name and public
// interface as radioDiags/WbFmDemodulator/WbFmDemodulator.h:23-31 of the reference,
// implemented over the C ABI of libhrfd.so (hrfd_demod_*, include/hrfd.h).
#ifndef HRFD_SHIM_WBFMDEMODULATOR_H
#define HRFD_SHIM_WBFMDEMODULATOR_H

#include "hrfd_shim_base.h"

class WbFmDemodulator : public hrfd_shim::DemodulatorBase
{
  public:

  WbFmDemodulator(void (*pcmCallbackPtr)(int16_t *bufferPtr,uint32_t bufferLength));
  ~WbFmDemodulator(void);

  void resetDemodulator(void);
  void setDemodulatorGain(float gain);
  void acceptIqData(int8_t *bufferPtr,uint32_t bufferLength);
  void displayInternalInformation(void);
};

#endif
