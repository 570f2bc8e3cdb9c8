// dropin_wiring.cc -- TEST INFRASTRUCTURE (oracle/Makefile target `dropin`, output oracle/_ref/dropin_app).
//
// What it proves: translation units of the REFERENCE application, compiled against the REFERENCE's OWN headers
// (hdr_diags/IqDataProcessor.h, BasebandDataProcessor.h, DataProvider.h, the (de)modulator headers: what Radio.cc
// sees through Radio.h:16-28, whatever the -I order), link against hackrfdiags_amd/csrc/shim/hrfd_shim.cc + libhrfd.so
// and run correctly -- the objects are allocated HERE with the reference's sizeof and used by the shim's member
// functions (layout containment, shim/hrfd_shim_layout.h).
//
// This file is the part of Radio.cc that cannot be linked in this image (Radio.cc needs libhackrf, which needs libusb):
// the object wiring of the Radio constructor (Radio.cc:164-237), the receive callback's hand-over to the data consumer
// (Radio.cc:3138-3164), the transmit callback's two sources (Radio.cc:3193-3244) and radioApp.cc's PCM sink
// (radioApp.cc:103-111).  Linked beside it, UNCHANGED and compiled where they lie: the reference's
// src_diags/DataConsumer.cc, MessageQueue.cc and UdpClient.cc.  No reference source is copied and nothing is stubbed.
//
//   dropin_app rx <mode 0..5> <n_blocks>          int8 IQ @ 2.048 MS/s on stdin -> int16 PCM on stdout
//   dropin_app tx <mode 0..5> <n_write> <n_read>  int16 PCM on stdin (the reference's reader thread takes it) -> int8 IQ
//   dropin_app file <path> <n_reads> <bytes>      DataProvider: `load iqfile` playback -> int8 IQ
//   dropin_app sizes                              sizeof of every class as THIS translation unit sees it
#include <stdarg.h>
#include <stdio.h>
#include <vector>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "IqDataProcessor.h"
#include "DataConsumer.h"
#include "DataProvider.h"
#include "AmDemodulator.h"
#include "FmDemodulator.h"
#include "WbFmDemodulator.h"
#include "SsbDemodulator.h"
#include "AmModulator.h"
#include "FmModulator.h"
#include "WbFmModulator.h"
#include "SsbModulator.h"
#include "BasebandDataProcessor.h"

// the two symbols the application provides (Radio.cc:15, diagUi.cc:2881)
uint32_t radio_adjustableReceiveGainInDb = 0;
void nprintf(FILE *s, const char *formatPtr, ...)
{
  va_list ap;
  va_start(ap, formatPtr);
  vfprintf(s, formatPtr, ap);
  va_end(ap);
}

static volatile unsigned pcmBlocksDelivered = 0;

static void processPcmData(int16_t *bufferPtr, uint32_t bufferLength)
{
  fwrite(bufferPtr, 2, bufferLength, stdout);
  pcmBlocksDelivered = pcmBlocksDelivered + 1;
}

static volatile unsigned magnitudeCallbacks = 0;
static void magnitudeSeen(uint32_t, void *) { magnitudeCallbacks = magnitudeCallbacks + 1; }

#define SHOW(T) fprintf(stderr, "sizeof %s %zu\n", #T, sizeof(T))

int main(int argc, char **argv)
{
  if (argc < 2) return 2;

  if (!strcmp(argv[1], "sizes"))
  {
    SHOW(IqDataProcessor); SHOW(BasebandDataProcessor); SHOW(DataProvider);
    SHOW(AmDemodulator); SHOW(FmDemodulator); SHOW(WbFmDemodulator); SHOW(SsbDemodulator);
    SHOW(AmModulator); SHOW(FmModulator); SHOW(WbFmModulator); SHOW(SsbModulator);
    return 0;
  }

  // ---- the Radio constructor's wiring, in its order (Radio.cc:164-237)
  char ip[] = "127.0.0.1";
  DataProvider *dataProviderPtr = new DataProvider();
  BasebandDataProcessor *transmitBasebandDataProcessorPtr = new BasebandDataProcessor();
  IqDataProcessor *receiveDataProcessorPtr = new IqDataProcessor(ip, 8001);
  DataConsumer *dataConsumerPtr = new DataConsumer(receiveDataProcessorPtr);
  AmDemodulator *amDemodulatorPtr = new AmDemodulator(processPcmData);
  receiveDataProcessorPtr->setAmDemodulator(amDemodulatorPtr);
  FmDemodulator *fmDemodulatorPtr = new FmDemodulator(processPcmData);
  receiveDataProcessorPtr->setFmDemodulator(fmDemodulatorPtr);
  WbFmDemodulator *wbFmDemodulatorPtr = new WbFmDemodulator(processPcmData);
  receiveDataProcessorPtr->setWbFmDemodulator(wbFmDemodulatorPtr);
  SsbDemodulator *ssbDemodulatorPtr = new SsbDemodulator(processPcmData);
  receiveDataProcessorPtr->setSsbDemodulator(ssbDemodulatorPtr);
  receiveDataProcessorPtr->setDemodulatorMode(IqDataProcessor::Fm);
  AmModulator *amModulatorPtr = new AmModulator();
  transmitBasebandDataProcessorPtr->setAmModulator(amModulatorPtr);
  FmModulator *fmModulatorPtr = new FmModulator();
  transmitBasebandDataProcessorPtr->setFmModulator(fmModulatorPtr);
  WbFmModulator *wbFmModulatorPtr = new WbFmModulator();
  transmitBasebandDataProcessorPtr->setWbFmModulator(wbFmModulatorPtr);
  SsbModulator *ssbModulatorPtr = new SsbModulator();
  transmitBasebandDataProcessorPtr->setSsbModulator(ssbModulatorPtr);
  transmitBasebandDataProcessorPtr->setModulatorMode(BasebandDataProcessor::None);

  int rc = 0;
  if (!strcmp(argv[1], "rx") && argc >= 4)
  {
    const int mode = atoi(argv[2]);
    const unsigned nBlocks = (unsigned)atoi(argv[3]);
    // Radio::setDemodulatorMode (Radio.cc:2396), then startReceiver's dataConsumerPtr->start()
    receiveDataProcessorPtr->setDemodulatorMode((IqDataProcessor::demodulatorType)mode);
    receiveDataProcessorPtr->registerSignalMagnitudeCallback(magnitudeSeen, NULL);   // what the AGC does (AutomaticGainControl.cc:45-70)
    receiveDataProcessorPtr->enableSignalMagnitudeNotification();
    dataConsumerPtr->start();
    // argv[4], when present: "a,b,c" -- the valid_length of consecutive USB transfers (hackRf/hackrf.c:1443: a transfer
    // that ends early is handed on with its actual_length; the last entry repeats).  DataConsumer::acceptData counts a
    // short one and passes it on, clips a long one (DataConsumer.cc:229-241).
    std::vector<uint32_t> lengths;
    for (const char *p = argc >= 5 ? argv[4] : ""; *p != 0;)
    {
      lengths.push_back((uint32_t)strtoul(p, NULL, 10));
      p = strchr(p, ',');
      if (p == NULL) break;
      p++;
    }
    uint32_t longest = DATA_CONSUMER_BUFFER_SIZE;
    for (uint32_t n : lengths) longest = n > longest ? n : longest;
    uint8_t *transfer = (uint8_t *)malloc(longest);
    uint32_t receiveTimeStamp = 0;
    for (unsigned b = 0; b < nBlocks; b++)
    {
      const uint32_t validLength = lengths.empty() ? DATA_CONSUMER_BUFFER_SIZE : lengths[b < lengths.size() ? b : lengths.size() - 1];
      if (fread(transfer, 1, validLength, stdin) != validLength) { rc = 3; break; }
      // Radio::receiveCallbackProcedure (Radio.cc:3138-3164), "USB thread" = this thread
      receiveTimeStamp += validLength >> 1;
      dataConsumerPtr->acceptData(receiveTimeStamp, transfer, validLength);
      // the radio delivers a block every 64 ms; here they come as fast as stdin gives them, so keep the 16-message
      // pool (DataConsumer.h:18) from being overrun: at most 8 blocks in front of the consumer thread
      for (int spin = 0; magnitudeCallbacks + 8 < b + 1 && spin < 20000; spin++) usleep(1000);
    }
    for (int spin = 0; magnitudeCallbacks < nBlocks && spin < 30000; spin++) usleep(1000);
    if (magnitudeCallbacks != nBlocks) rc = 4;
    dataConsumerPtr->stop();
    fflush(stdout);
    fprintf(stderr, "rx: %u blocks in, %u magnitude callbacks, %u PCM blocks out\n", nBlocks, magnitudeCallbacks,
            pcmBlocksDelivered);
    free(transfer);
  }
  else if (!strcmp(argv[1], "tx") && argc >= 5)
  {
    const int mode = atoi(argv[2]);
    const int nWrite = atoi(argv[3]), nRead = atoi(argv[4]);
    transmitBasebandDataProcessorPtr->setModulatorMode((BasebandDataProcessor::modulatorType)mode);
    // Radio::startLiveStream -> BasebandDataProcessor::start(): the reader thread takes stdin, 512 samples at a time
    transmitBasebandDataProcessorPtr->start();
    usleep(200000 + 10000 * (unsigned)nWrite);               // nWrite blocks are waiting in the pipe: all in the ring now
    int8_t *transfer = (int8_t *)malloc(262144);
    for (int r = 0; r < nRead; r++)
    {
      // Radio::transmitCallbackProcedure, case Live (Radio.cc:3221-3227)
      transmitBasebandDataProcessorPtr->getIqData(transfer, 262144);
      fwrite(transfer, 1, 262144, stdout);
    }
    transmitBasebandDataProcessorPtr->stop();
    fflush(stdout);
    free(transfer);
  }
  else if (!strcmp(argv[1], "file") && argc >= 5)
  {
    const int nReads = atoi(argv[3]);
    const uint32_t bytes = (uint32_t)atoi(argv[4]);
    if (!dataProviderPtr->loadIqFile(argv[2])) rc = 5;
    int8_t *transfer = (int8_t *)malloc(bytes);
    for (int r = 0; r < nReads && rc == 0; r++)
    {
      // Radio::transmitCallbackProcedure, case File (Radio.cc:3214-3219)
      dataProviderPtr->getIqData(transfer, bytes);
      fwrite(transfer, 1, bytes, stdout);
    }
    fflush(stdout);
    free(transfer);
  }
  else
  {
    rc = 2;
  }

  // Radio::~Radio deletes them (Radio.cc:263-330)
  delete dataConsumerPtr;
  delete receiveDataProcessorPtr;
  delete transmitBasebandDataProcessorPtr;
  delete dataProviderPtr;
  delete amDemodulatorPtr; delete fmDemodulatorPtr; delete wbFmDemodulatorPtr; delete ssbDemodulatorPtr;
  delete amModulatorPtr; delete fmModulatorPtr; delete wbFmModulatorPtr; delete ssbModulatorPtr;
  return rc;
}
