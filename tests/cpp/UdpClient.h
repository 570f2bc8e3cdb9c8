// tests/cpp/UdpClient.h -- TEST DOUBLE, not part of the product.
//
// The shim's IqDataProcessor sends the `enable iqdump` stream through the application's own UdpClient
// (radioDiags/hdr_diags/UdpClient.h; the application links its own UdpClient.o -- that class is outside the hot
// path, SURVEY section 2 row 24, and libhrfd ships no copy of it).  The shim tests and demos have no application to
// link against, so this header gives them a class with the two members the shim calls.  It only has to put the
// bytes on the wire the way the reference's wire format says: datagrams of at most 2048 bytes, in order.
#ifndef HRFD_TESTS_UDPCLIENT_DOUBLE_H
#define HRFD_TESTS_UDPCLIENT_DOUBLE_H

#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>

class UdpClient
{
  public:
  UdpClient(char *ip, int port) : fd_(::socket(AF_INET, SOCK_DGRAM, 0))
  {
    std::memset(&to_, 0, sizeof(to_));
    to_.sin_family = AF_INET;
    to_.sin_port = htons(static_cast<uint16_t>(port));
    ::inet_pton(AF_INET, ip, &to_.sin_addr);
  }
  ~UdpClient()
  {
    if (fd_ >= 0)
    {
      ::close(fd_);
    }
  }
  UdpClient(const UdpClient &) = delete;
  UdpClient &operator=(const UdpClient &) = delete;

  bool connectionIsEstablished() { return fd_ >= 0; }

  // true when every byte went out
  bool sendData(void *data, int length)
  {
    const char *p = static_cast<const char *>(data);
    bool all = fd_ >= 0;
    for (int off = 0; off < length && fd_ >= 0; off += kDatagram)
    {
      const int n = std::min(kDatagram, length - off);
      all = (::sendto(fd_, p + off, static_cast<size_t>(n), 0, reinterpret_cast<const sockaddr *>(&to_), sizeof(to_)) == n) && all;
    }
    return all;
  }

  private:
  static constexpr int kDatagram = 2048;
  int fd_;
  sockaddr_in to_;
};

#endif
