// SsbModulator.h -- drop-in replacement header: same class name and public interface
// as radioDiags/SsbModulator/SsbModulator.h:23-35 of the reference, implemented over
// the C ABI of libhrfd.so (hrfd_mod_*, include/hrfd.h).
#ifndef HRFD_SHIM_SSBMODULATOR_H
#define HRFD_SHIM_SSBMODULATOR_H

#include <stdint.h>

#include "hrfd.h"

class SsbModulator
{
  public:

  SsbModulator(void);
  ~SsbModulator(void);

  void resetModulator(void);
  void setLsbModulationMode(void);
  void setUsbModulationMode(void);

  void acceptData(int16_t *bufferPtr,
                  uint32_t bufferLength,
                  int8_t *outputBufferPtr,
                  uint32_t *outputBufferLengthPtr);

  void displayInternalInformation(void);

  private:

  bool lsbModulationMode;
  hrfd_mod *handle;
};

#endif
