// DataProvider.h -- drop-in replacement header: same class name and public interface as
// radioDiags/hdr_diags/DataProvider.h:14-26 of the reference (`load iqfile`: cyclic playback of
// a raw int8 IQ file).  The file image lives in HBM (hrfd_play_*); getIqData copies the next
// bufferLength bytes, wrapping at the end of the file.
#ifndef HRFD_SHIM_DATAPROVIDER_H
#define HRFD_SHIM_DATAPROVIDER_H

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
