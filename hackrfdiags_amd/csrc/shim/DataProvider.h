// DataProvider.h -- drop-in replacement header: same class name and public interface as
// radioDiags/hdr_diags/DataProvider.h:14-26 of the reference (`load iqfile`: cyclic playback of
// a raw int8 IQ file).  The file image lives in HBM (hrfd_play_*); getIqData copies the next
// bufferLength bytes, wrapping at the end of the file.
// The include guard is the REFERENCE header's own: in a translation unit that has already seen the reference's
// declaration of this class (Radio.h includes its neighbours by quoted name) this header must be a no-op, and the
// other way round; the two declarations are interchangeable by construction (hrfd_shim_layout.h).
#ifndef _DATAPROVIDER_H_
#define _DATAPROVIDER_H_
#define HRFD_SHIM_DECLARES_DATAPROVIDER 1

#include <stdint.h>

#include "hrfd.h"

class DataProvider
{
  public:

  DataProvider(void);
  ~DataProvider(void);

  void getIqData(int8_t *bufferPtr,uint32_t bufferLength);
  bool loadIqFile(char *fileNamePtr);
  void displayInternalInformation(void);

  private:

  hrfd_play *handle;
  char iqFileName[256];
};

#endif
