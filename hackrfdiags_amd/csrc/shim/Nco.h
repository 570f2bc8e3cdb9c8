// Nco.h -- drop-in replacement header: same class name and public interface as
// radioDiags/Nco/Nco.h:19-29 of the reference, implemented over hrfd_nco_*.
// One sample per call crosses the PCIe bus here; the batched entry point
// hrfd_nco_run(count) is what a GPU-resident modulator should use.
#ifndef HRFD_SHIM_NCO_H
#define HRFD_SHIM_NCO_H

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
