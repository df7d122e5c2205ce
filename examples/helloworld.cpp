// helloworld -- the reference's examples/helloworld/helloworld.go against the C++ mirror of package
// mpi (BASELINE config 1).  Same behaviour: every rank prints its greeting, then concurrently
// sends a string to every rank (itself included) with tag 0 and receives one from every rank.
//   xmpirun N helloworld [--tcp]   (or: helloworld -mpi-addr :6000 -mpi-alladdr :6000,:6001,...)
#include <cstdio>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <cstring>

#include "mpi.hpp"
#include "network.hpp"

int main(int argc, char** argv) {
  mpi::ParseFlags(&argc, argv);  // flag.Parse() must be called to set the addresses (mpi.go:43)
  // --tcp: the reference's own backend (TCP + gob, wire-compatible: the other ranks may be reference processes)
  // instead of the xGMI one -- what `mpi.Register(&mpi.Network{})` is in a reference program (mpi.go:56-67)
  static mpi::Network net;
  for (int i = 1; i < argc; i++)
    if (!strcmp(argv[i], "--tcp")) mpi::Register(&net);
  if (mpi::Error err = mpi::Init()) {
    fprintf(stderr, "%s\n", err.What().c_str());
    return 1;
  }
  const int rank = mpi::Rank();
  if (rank == -1) {
    fprintf(stderr, "Incorrect initialization\n");
    return 1;
  }
  const int size = mpi::Size();
  printf("Hello world, I'm node %d in a land with %d nodes\n", rank, size);
  fflush(stdout);

  std::vector<std::thread> workers;  // one per goroutine of the reference
  std::mutex out;
  int failures = 0;
  for (int i = 0; i < size; i++)
    workers.emplace_back([&, i] {
      char text[128];
      if (i == rank) snprintf(text, sizeof text, "\"I'm just node %d talking to myself\"", rank);
      else snprintf(text, sizeof text, "\"Hello node %d, I'm node %d\"", i, rank);
      const std::string str = text;
      if (mpi::Error err = mpi::Send(mpi::Slice(str), i, 0)) {
        std::lock_guard<std::mutex> g(out);
        fprintf(stderr, "%s\n", err.What().c_str());
        failures++;
      }
    });
  for (int i = 0; i < size; i++)
    workers.emplace_back([&, i] {
      std::string str;
      mpi::Error err = mpi::Receive(mpi::Into(&str), i, 0);
      std::lock_guard<std::mutex> g(out);
      if (err) {
        fprintf(stderr, "%s\n", err.What().c_str());
        failures++;
      } else {
        printf("I, node %d, received a message: %s\n", rank, str.c_str());
      }
    });
  for (auto& t : workers) t.join();
  fflush(stdout);
  mpi::Finalize();
  return failures ? 1 : 0;
}
