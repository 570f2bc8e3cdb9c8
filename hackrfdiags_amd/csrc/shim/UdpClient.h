// UdpClient.h -- drop-in replacement header: same class name and public interface as
// radioDiags/hdr_diags/UdpClient.h:13-31 of the reference.  The wire format of `enable iqdump`:
// raw interleaved int8 IQ at 256 kS/s in datagrams of at most 2048 bytes (UdpClient.cc:38,173-241).
#ifndef HRFD_SHIM_UDPCLIENT_H
#define HRFD_SHIM_UDPCLIENT_H

#include <unistd.h>
#include <sys/types.h>
#include <sys/socket.h>
#include <netinet/in.h>
#include <arpa/inet.h>
#include <string.h>

class UdpClient
{
  public:

  UdpClient(char *ipAddressPtr,int port);
  ~UdpClient(void);

  bool connectionIsEstablished(void);
  bool sendData(void *bufferPtr,int bufferLength);

  private:

  size_t maxPayloadLength;
  int socketDescriptor;
  struct sockaddr_in peerAddress;
};

#endif
