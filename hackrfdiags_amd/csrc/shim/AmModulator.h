// AmModulator.h -- drop-in replacement header: same class name and public interface
// as radioDiags/AmModulator/AmModulator.h:20-33 of the reference, implemented over
// the C ABI of libhrfd.so (hrfd_mod_*, include/hrfd.h).
#ifndef HRFD_SHIM_AMMODULATOR_H
#define HRFD_SHIM_AMMODULATOR_H

#include <stdint.h>

#include "hrfd.h"

class AmModulator
{
  public:

  AmModulator(void);
  ~AmModulator(void);

  void resetModulator(void);
  void setModulationIndex(float modulationIndex);

  void acceptData(int16_t *bufferPtr,
                  uint32_t bufferLength,
                  int8_t *outputBufferPtr,
                  uint32_t *outputBufferLengthPtr);

  void displayInternalInformation(void);

  private:

  float modulationIndex;
  hrfd_mod *handle;
};

#endif
