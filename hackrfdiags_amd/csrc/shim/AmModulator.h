// AmModulator.h -- drop-in replacement header: same class name and public interface
// as radioDiags/AmModulator/AmModulator.h:20-33 of the reference, implemented over
// the C ABI of libhrfd.so (hrfd_mod_*, include/hrfd.h).
// The include guard is the REFERENCE header's own: in a translation unit that has already seen the reference's
// declaration of this class (Radio.h includes its neighbours by quoted name) this header must be a no-op, and the
// other way round; the two declarations are interchangeable by construction (hrfd_shim_layout.h).
#ifndef __AMMODULATOR__
#define __AMMODULATOR__
#define HRFD_SHIM_DECLARES_AMMODULATOR 1

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
