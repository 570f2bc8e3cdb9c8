// FmDemodulator.h -- drop-in replacement header: same class name and public
// interface as radioDiags/FmDemodulator/FmDemodulator.h:23-31 of the reference,
// implemented over the C ABI of libhrfd.so (hrfd_demod_*, include/hrfd.h).
// The include guard is the REFERENCE header's own: in a translation unit that has already seen the reference's
// declaration of this class (Radio.h includes its neighbours by quoted name) this header must be a no-op, and the
// other way round; the two declarations are interchangeable by construction (hrfd_shim_layout.h).
#ifndef __FMDEMODULATOR__
#define __FMDEMODULATOR__
#define HRFD_SHIM_DECLARES_FMDEMODULATOR 1

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
