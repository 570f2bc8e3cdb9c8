// WbFmModulator.h -- drop-in replacement header: same class name and public interface
// as radioDiags/WbFmModulator/WbFmModulator.h:20-33 of the reference, implemented over
// the C ABI of libhrfd.so (hrfd_mod_*, include/hrfd.h).
#ifndef HRFD_SHIM_WBFMMODULATOR_H
#define HRFD_SHIM_WBFMMODULATOR_H

#include <stdint.h>

#include "hrfd.h"

class WbFmModulator
{
  public:

  WbFmModulator(void);
  ~WbFmModulator(void);

  void resetModulator(void);
  void setFrequencyDeviation(float deviaton);

  void acceptData(int16_t *bufferPtr,
                  uint32_t bufferLength,
                  int8_t *outputBufferPtr,
                  uint32_t *outputBufferLengthPtr);

  void displayInternalInformation(void);

  private:

  float frequencyDeviation;
  hrfd_mod *handle;
};

#endif
