// WbFmModulator.h -- drop-in replacement header: same class name and public interface
// as radioDiags/WbFmModulator/WbFmModulator.h:20-33 of the reference, implemented over
// the C ABI of libhrfd.so (hrfd_mod_*, include/hrfd.h).
// The include guard is the REFERENCE header's own: in a translation unit that has already seen the reference's
// declaration of this class (Radio.h includes its neighbours by quoted name) this header must be a no-op, and the
// other way round; the two declarations are interchangeable by construction (hrfd_shim_layout.h).
#ifndef __WBFMMODULATOR__
#define __WBFMMODULATOR__
#define HRFD_SHIM_DECLARES_WBFMMODULATOR 1

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
