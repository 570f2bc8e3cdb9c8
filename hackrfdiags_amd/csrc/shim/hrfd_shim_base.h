// hrfd_shim_base.h -- what the reference-named shim classes share.  Not part of
// the reference's interface; kept in a namespace of its own.
#ifndef HRFD_SHIM_BASE_H
#define HRFD_SHIM_BASE_H

#include <stdint.h>
#include <stdio.h>

#include "hrfd.h"

class IqDataProcessor;

namespace hrfd_shim {

// One demodulator instance = one channel of an hrfd_demod handle (inner boundary).
// The handle is created lazily so that objects can be constructed on machines
// without a GPU (as the reference constructs all four demodulators up front);
// the first acceptIqData call fails loudly when there is no device.
class DemodulatorBase
{
  public:

  float currentGain(void) const { return gain; }

  protected:

  DemodulatorBase(int mode, float defaultGain,
                  void (*pcmCallbackPtr)(int16_t *bufferPtr,uint32_t bufferLength));
  ~DemodulatorBase(void);

  void reset(void);
  void setGain(float gain);
  void setSideband(bool lsb);
  void accept(int8_t *bufferPtr,uint32_t bufferLength);
  void display(const char *name);

  // the outer boundary (IqDataProcessor shim) hands the PCM of the fused kernel
  // to the registered demodulator's callback through this
  void deliverPcm(int16_t *bufferPtr,uint32_t bufferLength);
  friend class ::IqDataProcessor;

  int mode;
  float gain;
  hrfd_demod *handle;
  void (*pcmCallbackPtr)(int16_t *bufferPtr,uint32_t bufferLength);
  int16_t pcmData[512];
};

void fatal(const char *what, int rc);
uint32_t sizeClip(const char *what, unsigned long byteCount, uint32_t limit);

} // namespace hrfd_shim

#endif
