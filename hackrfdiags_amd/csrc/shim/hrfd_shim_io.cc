// hrfd_shim_io.cc -- the reference-named host I/O classes around the hot path that need no GPU:
// UdpClient (the `enable iqdump` wire format).  Plain host C++, no libhrfd dependency.
#include <stdio.h>

#include "UdpClient.h"

// UdpClient.cc:30-72: a datagram socket towards ipAddress:port, 2048-byte payloads ("this size
// interoperates with netcat"), 32768 bytes of send buffer
UdpClient::UdpClient(char *ipAddressPtr,int port)
{
  int bufferLength = 32768;
  maxPayloadLength = 2048;
  socketDescriptor = socket(PF_INET, SOCK_DGRAM, 0);
  if (socketDescriptor != -1)
  {
    memset(&peerAddress, 0, sizeof(peerAddress));
    peerAddress.sin_family = AF_INET;
    peerAddress.sin_addr.s_addr = inet_addr(ipAddressPtr);
    peerAddress.sin_port = htons((uint16_t)port);
    (void)setsockopt(socketDescriptor, SOL_SOCKET, SO_SNDBUF, &bufferLength, sizeof(bufferLength));
  }
  else
  {
    socketDescriptor = 0;                                // UdpClient.cc: 0 means "no socket"
  }
}

UdpClient::~UdpClient(void)
{
  if (socketDescriptor != 0)
  {
    close(socketDescriptor);
  }
}

bool UdpClient::connectionIsEstablished(void)
{
  return (socketDescriptor != 0);
}

// UdpClient.cc:173-241: bufferLength / 2048 full datagrams, then the remainder if there is one.
// The reference returns its failure flag (success = failureOccurred); kept, callers ignore it
// (IqDataProcessor.cc:956).
bool UdpClient::sendData(void *bufferPtr,int bufferLength)
{
  bool failureOccurred = false;
  bool success = false;
  unsigned char *octetPtr = (unsigned char *)bufferPtr;
  const int numberOfBlocks = bufferLength / (int)maxPayloadLength;
  const int remainder = bufferLength % (int)maxPayloadLength;

  if (socketDescriptor != 0)
  {
    for (int i = 0; i < numberOfBlocks; i++)
    {
      const ssize_t count = sendto(socketDescriptor, octetPtr, maxPayloadLength, 0,
                                   (struct sockaddr *)&peerAddress, sizeof(struct sockaddr));
      octetPtr += maxPayloadLength;
      if (count != (ssize_t)maxPayloadLength)
      {
        failureOccurred = true;
      }
    }
    if (remainder != 0)
    {
      const ssize_t count = sendto(socketDescriptor, octetPtr, (size_t)remainder, 0,
                                   (struct sockaddr *)&peerAddress, sizeof(struct sockaddr));
      if (count != remainder)
      {
        failureOccurred = true;
      }
    }
    success = failureOccurred;
  }
  return success;
}
