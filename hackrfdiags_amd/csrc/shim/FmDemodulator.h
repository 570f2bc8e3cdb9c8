// FmDemodulator.h -- drop-in replacement header: same class name and public
// interface as radioDiags/FmDemodulator/FmDemodulator.h:23-31 of the reference,
// implemented over the C ABI of libhrfd.so (hrfd_demod_*, include/hrfd.h).
#ifndef HRFD_SHIM_FMDEMODULATOR_H
#define HRFD_SHIM_FMDEMODULATOR_H

#include "hrfd_shim_base.h"

class FmDemodulator : public hrfd_shim::DemodulatorBase
{
  public:

  FmDemodulator(void (*pcmCallbackPtr)(int16_t *bufferPtr,uint32_t bufferLength));
  ~FmDemodulator(void);

  void resetDemodulator(void);
  void setDemodulatorGain(float gain);
  void acceptIqData(int8_t *bufferPtr,uint32_t bufferLength);
  void displayInternalInformation(void);
};

#endif
