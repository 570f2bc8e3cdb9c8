// Nco.h -- drop-in replacement header: same class name and public interface as
// radioDiags/Nco/Nco.h:19-29 of the reference, implemented over hrfd_nco_*.
// One sample per call crosses the PCIe bus here; the batched entry point
// hrfd_nco_run(count) is what a GPU-resident modulator should use.
// The include guard is the REFERENCE header's own: in a translation unit that has already seen the reference's
// declaration of this class (Radio.h includes its neighbours by quoted name) this header must be a no-op, and the
// other way round; the two declarations are interchangeable by construction (hrfd_shim_layout.h).
#ifndef __NCO__
#define __NCO__
#define HRFD_SHIM_DECLARES_NCO 1

#include <stdint.h>

#include "hrfd.h"

class Nco
{
  public:

  Nco(float sampleRate,float frequency);
  ~Nco(void);

  void setFrequency(float frequency);
  void reset(void);
  void run(float *iValuePtr,float *qValuePtr);
  void runFast(float *iValuePtr,float *qValuePtr);

  private:

  float sampleRate;
  float frequency;
  hrfd_nco *handle;
};

#endif
