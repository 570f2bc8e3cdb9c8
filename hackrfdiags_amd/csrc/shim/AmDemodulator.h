// AmDemodulator.h -- drop-in replacement header: same class name and public
// interface as radioDiags/AmDemodulator/AmDemodulator.h:23-31 of the reference,
// implemented over the C ABI of libhrfd.so (hrfd_demod_*, include/hrfd.h).
#ifndef HRFD_SHIM_AMDEMODULATOR_H
#define HRFD_SHIM_AMDEMODULATOR_H

#include "hrfd_shim_base.h"

class AmDemodulator : public hrfd_shim::DemodulatorBase
{
  public:

  AmDemodulator(void (*pcmCallbackPtr)(int16_t *bufferPtr,uint32_t bufferLength));
  ~AmDemodulator(void);

  void resetDemodulator(void);
  void setDemodulatorGain(float gain);
  void acceptIqData(int8_t *bufferPtr,uint32_t bufferLength);
  void displayInternalInformation(void);
};

#endif
