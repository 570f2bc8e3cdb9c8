// tests/cpp/shim_demo.cc -- a miniature of the reference application's receive wiring
// (radioApp.cc:103-111,238-242; Radio.cc:164-203; DataConsumer.cc:341) on top of the
// drop-in shim classes: blocks of int8 IQ on stdin -> PCM on stdout, driven through
//   IqDataProcessor::acceptIqData            (outer boundary, argv[2] == "outer")
//   XDemodulator::acceptIqData on 256 kS/s   (inner boundary, argv[2] == "inner")
// usage: shim_demo <mode 1..5> <outer|inner> <block_bytes> [iqdump udp port]
//        (block_bytes may be a comma-separated list -- "1000,262144": the byte counts of consecutive calls, the last one
//         repeating -- for short USB transfers; with HRFD_DEMO_COUNTS=1 every PCM callback also prints "pcm <count>" on
//         stderr, a callback with nothing in it included)
//        shim_demo <port> udp <bytes per sendData>            (stdin -> UdpClient datagrams)
//        shim_demo <calls> provider <bytes per call> <file>   (DataProvider playback -> stdout)
//        shim_demo <mode 0..5> bbp 0 <schedule of w/r/s/p>     (BasebandDataProcessor: PCM blocks on stdin through the ring
//                                                             and the mode's modulator, 262144 bytes of IQ per block -> stdout)
//        shim_demo <0|1> fs4 <bytes>                          (upconvertByFsOver4 / downconvertByFsOver4 on stdin -> stdout)
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <vector>

#include "IqDataProcessor.h"
#include "SsbModulator.h"
#include "AmModulator.h"
#include "FmModulator.h"
#include "WbFmModulator.h"
#include "DataProvider.h"
#include "BasebandDataProcessor.h"

uint32_t radio_adjustableReceiveGainInDb = 0;          // Radio.cc:15
void nprintf(FILE *s, const char *formatPtr, ...)      // diagUi.cc:2881
{
  va_list ap;
  va_start(ap, formatPtr);
  vfprintf(s, formatPtr, ap);
  va_end(ap);
}

static bool g_counts = false;
static void processPcmData(int16_t *bufferPtr, uint32_t bufferLength)   // radioApp.cc:103
{
  fwrite(bufferPtr, sizeof(int16_t), bufferLength, stdout);
  if (g_counts)
  {
    fprintf(stderr, "pcm %u\n", bufferLength);
  }
}

int main(int argc, char **argv)
{
  if (argc < 4)
  {
    fprintf(stderr, "usage: %s <mode> <outer|inner|ssbmod|ammod|fmmod|wbfmmod> <block_bytes>\n", argv[0]);
    return 2;
  }
  const int mode = atoi(argv[1]);
  const bool outer = strcmp(argv[2], "outer") == 0;
  size_t blockBytes = (size_t)atoi(argv[3]);
  std::vector<size_t> sizes;                                   // "a,b,c": the byte counts of consecutive calls
  for (const char *p = argv[3]; *p != 0;)
  {
    sizes.push_back((size_t)strtoul(p, NULL, 10));
    p = strchr(p, ',');
    if (p == NULL) break;
    p++;
  }
  size_t longest = blockBytes;
  for (size_t n : sizes) longest = n > longest ? n : longest;
  std::vector<int8_t> buf(longest + 2);                        // (an odd count reads one byte further, as the reference does)
  g_counts = getenv("HRFD_DEMO_COUNTS") != NULL;

  if (strcmp(argv[2], "ssbmod") == 0)
  {
    // transmit: 512 PCM samples per call -> 262144 bytes of IQ (BasebandDataProcessor.cc:682)
    SsbModulator mod;
    if (mode == 5) mod.setUsbModulationMode();
    std::vector<int16_t> pcm(512);
    std::vector<int8_t> iq(262144);
    while (fread(pcm.data(), 2, 512, stdin) == 512)
    {
      uint32_t outBytes = 0;
      mod.acceptData(pcm.data(), 512, iq.data(), &outBytes);
      fwrite(iq.data(), 1, outBytes, stdout);
    }
    return 0;
  }

  if (strcmp(argv[2], "ammod") == 0 || strcmp(argv[2], "fmmod") == 0 || strcmp(argv[2], "wbfmmod") == 0)
  {
    // the other two modulators behind the same call (BasebandDataProcessor.cc:660-697); <mode> is
    // the setter's argument in thousandths when non-zero (modulation index / deviation in Hz)
    const bool am = strcmp(argv[2], "ammod") == 0;
    const bool wb = strcmp(argv[2], "wbfmmod") == 0;
    AmModulator amMod;
    FmModulator fmMod;
    WbFmModulator wbMod;
    if (mode != 0)
    {
      if (am) amMod.setModulationIndex((float)mode / 1000);
      else if (wb) wbMod.setFrequencyDeviation((float)mode);
      else fmMod.setFrequencyDeviation((float)mode);
    }
    std::vector<int16_t> pcm(512);
    std::vector<int8_t> iq(262144);
    while (fread(pcm.data(), 2, 512, stdin) == 512)
    {
      uint32_t outBytes = 0;
      if (am) amMod.acceptData(pcm.data(), 512, iq.data(), &outBytes);
      else if (wb) wbMod.acceptData(pcm.data(), 512, iq.data(), &outBytes);
      else fmMod.acceptData(pcm.data(), 512, iq.data(), &outBytes);
      fwrite(iq.data(), 1, outBytes, stdout);
    }
    return 0;
  }

  if (strcmp(argv[2], "bbp") == 0)
  {
    // transmit boundary (BasebandDataProcessor.cc:381, 630-697): the ring is filled by hand (putPcmBlock) instead of the
    // stdin reader thread so that the pacing is deterministic: the stream runs, seven blocks of lead, then one block
    // in, one transfer buffer out -- the lag stays between the "repeat" and the "drop" marks
    BasebandDataProcessor bbp;
    AmModulator am;
    FmModulator fm;
    WbFmModulator wb;
    SsbModulator ssb;
    bbp.setAmModulator(&am);
    bbp.setFmModulator(&fm);
    bbp.setWbFmModulator(&wb);
    bbp.setSsbModulator(&ssb);
    bbp.setModulatorMode((BasebandDataProcessor::modulatorType)mode);
    // argv[4]: a schedule -- 'w' one PCM block from stdin into the ring, 'r' one transfer buffer out,
    // 's' / 'p' the stream starts (without the stdin reader thread) / stops
    const char *ops = argc >= 5 ? argv[4] : "r";
    std::vector<int16_t> pcm(512);
    std::vector<int8_t> iq(262144);
    for (const char *o = ops; *o; o++)
    {
      if (*o == 'w')
      {
        if (fread(pcm.data(), 2, 512, stdin) != 512) return 4;
        bbp.putPcmBlock(pcm.data());
      }
      else if (*o == 'r')
      {
        bbp.getIqData(iq.data(), 262144);
        fwrite(iq.data(), 1, 262144, stdout);
      }
      else if (*o == 's') bbp.startWithoutReader();
      else if (*o == 'p') bbp.stop();
    }
    bbp.displayInternalInformation();
    return 0;
  }
  if (strcmp(argv[2], "fs4") == 0)
  {
    IqDataProcessor *p = NULL;
    static char ip2[] = "127.0.0.1";
    IqDataProcessor proc2(ip2, 8001);
    p = &proc2;
    const size_t got = fread(buf.data(), 1, blockBytes, stdin);
    if (mode) p->upconvertByFsOver4(buf.data(), (uint32_t)got);
    else p->downconvertByFsOver4(buf.data(), (uint32_t)got);
    fwrite(buf.data(), 1, got, stdout);
    return 0;
  }
  if (strcmp(argv[2], "udp") == 0)
  {
    // UdpClient alone: stdin -> datagrams towards 127.0.0.1:<mode>, <block_bytes> per sendData call
    static char ip[] = "127.0.0.1";
    UdpClient client(ip, mode);
    size_t got;
    while ((got = fread(buf.data(), 1, blockBytes, stdin)) > 0)
    {
      client.sendData(buf.data(), (int)got);
    }
    return 0;
  }
  if (strcmp(argv[2], "provider") == 0)
  {
    // `load iqfile` playback: DataProvider::getIqData, <mode> calls of <block_bytes>, file = argv[4]
    DataProvider provider;
    if (argc < 5 || !provider.loadIqFile(argv[4]))
    {
      fprintf(stderr, "cannot load the iq file\n");
      return 3;
    }
    for (int i = 0; i < mode; i++)
    {
      provider.getIqData(buf.data(), (uint32_t)blockBytes);
      fwrite(buf.data(), 1, blockBytes, stdout);
    }
    provider.displayInternalInformation();
    return 0;
  }

  static char ip[] = "127.0.0.1";
  // argv[4], when present: the UDP port of the `enable iqdump` stream (IqDataProcessor.cc:953-957)
  IqDataProcessor proc(ip, argc >= 5 ? atoi(argv[4]) : 8001);
  if (argc >= 5)
  {
    proc.enableIqDump();
  }
  AmDemodulator am(processPcmData);
  FmDemodulator fm(processPcmData);
  WbFmDemodulator wbfm(processPcmData);
  SsbDemodulator ssb(processPcmData);
  proc.setAmDemodulator(&am);
  proc.setFmDemodulator(&fm);
  proc.setWbFmDemodulator(&wbfm);
  proc.setSsbDemodulator(&ssb);
  proc.setDemodulatorMode((IqDataProcessor::demodulatorType)mode);

  unsigned long timeStamp = 0;
  size_t call = 0;
  while (true)
  {
    blockBytes = sizes[call < sizes.size() ? call : sizes.size() - 1];
    call++;
    if (fread(buf.data(), 1, blockBytes, stdin) != blockBytes)
    {
      break;
    }
    buf[blockBytes] = 0;                                       // what an odd count reads one byte further
    if (outer)
    {
      proc.acceptIqData(timeStamp++, buf.data(), blockBytes);
    }
    else
    {
      switch (mode)
      {
        case 1: am.acceptIqData(buf.data(), (uint32_t)blockBytes); break;
        case 2: fm.acceptIqData(buf.data(), (uint32_t)blockBytes); break;
        case 3: wbfm.acceptIqData(buf.data(), (uint32_t)blockBytes); break;
        default: ssb.acceptIqData(buf.data(), (uint32_t)blockBytes); break;
      }
    }
  }
  proc.displayInternalInformation();
  return 0;
}
