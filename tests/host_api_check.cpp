// Host-logic checks of the C++ mirror of package mpi (no GPU needed): flag parsing (flags.go:44-50),
// the "not initialised" answers (mpi.go:110-111, network.go:41-50), Register's once-only rule
// (mpi.go:61-67) and rank assignment errors (network.go:94-109).  Prints "ok" and exits 0.
#include <cstdio>
#include <cstring>
#include <stdexcept>

#include "mpi.hpp"

#define CHECK(c)                                              \
  do {                                                        \
    if (!(c)) {                                               \
      fprintf(stderr, "check failed: %s (line %d)\n", #c, __LINE__); \
      return 1;                                               \
    }                                                         \
  } while (0)

struct Fake : mpi::Interface {
  int inits = 0;
  mpi::Error Init() override { inits++; return mpi::Error(); }
  void Finalize() override {}
  int Rank() override { return 3; }
  int Size() override { return 5; }
  mpi::Error Send(const mpi::Data&, int, int) override { return mpi::Error(XMPI_ERR_TAG_EXISTS, "Tag 7 already in use sending"); }
  mpi::Error Receive(mpi::Data, int, int) override { return mpi::Error(); }
};

int main() {
  // before Init: Rank() == -1, Size() == 0
  CHECK(mpi::Rank() == -1);
  CHECK(mpi::Size() == 0);

  const char* args[] = {"prog", "-x", "-mpi-addr", ":6001", "--mpi-alladdr=:6002,:6000,:6001", "keep", "-mpi-inittimeout", "1m30s",
                        "-mpi-password=pw", "-mpi-protocol", "tcp"};
  int argc = (int)(sizeof args / sizeof args[0]);
  char* argv[16];
  for (int i = 0; i < argc; i++) argv[i] = const_cast<char*>(args[i]);
  mpi::ParseFlags(&argc, argv);
  CHECK(argc == 3 && !strcmp(argv[1], "-x") && !strcmp(argv[2], "keep"));
  CHECK(mpi::FlagAddr == ":6001");
  CHECK(mpi::FlagAllAddrs.size() == 3 && mpi::FlagAllAddrs[1] == ":6000");
  CHECK(mpi::FlagInitTimeout == 90.0);
  CHECK(mpi::FlagPassword == "pw" && mpi::FlagProtocol == "tcp");

  // address not in the list / duplicate addresses are Init errors (network.go:97-105)
  {
    mpi::XGMI x;
    x.Addr = ":7000";
    x.Addrs = {":6000", ":6001"};
    mpi::Error e = x.Init();
    CHECK(e && e.What().find("local ip address not in global list") != std::string::npos);
    mpi::XGMI y;
    y.Addr = ":6000";
    y.Addrs = {":6000", ":6000"};
    e = y.Init();
    CHECK(e && e.What().find("not unique") != std::string::npos);
  }

  // Register swaps the backend behind the package-level functions; a second call "panics"
  Fake fake;
  mpi::Register(&fake);
  CHECK(!mpi::Init() && fake.inits == 1);
  CHECK(mpi::Rank() == 3 && mpi::Size() == 5);
  mpi::Error e = mpi::Send(mpi::Slice(std::string("x")), 1, 7);
  CHECK(e.IsTagExists() && e.What() == "Tag 7 already in use sending");
  CHECK(mpi::Allreduce(mpi::Data(), mpi::Data()).Code() == XMPI_ERR_UNSUPPORTED);  // Fake has no collectives
  bool threw = false;
  try {
    mpi::Register(&fake);
  } catch (const std::logic_error& ex) {
    threw = !strcmp(ex.what(), "register called more than once");
  }
  CHECK(threw);
  printf("ok\n");
  return 0;
}
