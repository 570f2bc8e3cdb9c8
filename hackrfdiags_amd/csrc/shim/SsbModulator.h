// SsbModulator.h -- drop-in replacement header: same class name and public interface
// as radioDiags/SsbModulator/SsbModulator.h:23-35 of the reference, implemented over
// the C ABI of libhrfd.so (hrfd_mod_*, include/hrfd.h).
// The include guard is the REFERENCE header's own: in a translation unit that has already seen the reference's
// declaration of this class (Radio.h includes its neighbours by quoted name) this header must be a no-op, and the
// other way round; the two declarations are interchangeable by construction (hrfd_shim_layout.h).
#ifndef __SSBMODULATOR__
#define __SSBMODULATOR__
#define HRFD_SHIM_DECLARES_SSBMODULATOR 1

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
