// tests/cpp/udp_demo.cc -- the UdpClient test double alone (tests/cpp/UdpClient.h; no libhrfd): stdin -> datagrams towards
// 127.0.0.1:<port>, <bytes> per sendData call.   usage: udp_demo <port> <bytes per call>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "UdpClient.h"

int main(int argc, char **argv)
{
  if (argc < 3)
  {
    return 2;
  }
  static char ip[] = "127.0.0.1";
  UdpClient client(ip, atoi(argv[1]));
  std::vector<unsigned char> buf((size_t)atoi(argv[2]));
  size_t got;
  while ((got = fread(buf.data(), 1, buf.size(), stdin)) > 0)
  {
    client.sendData(buf.data(), (int)got);
  }
  return client.connectionIsEstablished() ? 0 : 1;
}
