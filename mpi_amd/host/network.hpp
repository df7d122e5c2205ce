// network.hpp -- mpi::Network: the reference's TCP backend (type Network, network.go:25-39) for the C++ mirror of
// package mpi, wire-compatible with the reference (gob framing, message + ack; see network.cpp, gobwire.hpp).
// Register it like the reference's programs would register a backend:
//     static mpi::Network net;  mpi::Register(&net);      // mpi.go:61-67
// It is the CPU path -- host slices over sockets -- kept for jobs that mix ranks with and without a GPU, for
// multi-node runs, and as the peer that checks this project's reading of the reference's protocol.
#pragma once
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "mpi.hpp"

namespace mpi {

// It also provides the collectives the reference only stubs (mpi.go:130), composed the way a reference user would
// compose them -- every rank exchanges whole buffers with Send / Receive (the all-to-all idiom of
// examples/helloworld/helloworld.go:53-81) and folds them on the host in rank order 0..N-1, one rounding per
// operation.  That composition is, word for word, what this repository's oracle DEFINES the collectives to be
// (DESIGN.md section 5); here it is a backend a program can run, so the xGMI backend's results can be compared with
// it bit for bit on the same inputs.
class Network : public Interface, public Collective {
 public:
  // the reference's fields (network.go:25-39); zero values are filled from the flags (network.go:69-90)
  std::string NetProto;            // -mpi-protocol; "tcp"
  std::string Addr;                // -mpi-addr
  std::vector<std::string> Addrs;  // -mpi-alladdr
  double Timeout = 0;              // -mpi-inittimeout, seconds; 0 = keep trying
  std::string Password;            // -mpi-password

  ~Network() override;
  Error Init() override;
  void Finalize() override;
  int Rank() override;
  int Size() override;
  Error Send(const Data& data, int destination, int tag) override;
  Error Receive(Data data, int source, int tag) override;

  Error Bcast(Data buf, int root) override;
  Error Reduce(const Data& send, Data recv, xmpi_op op, int root) override;
  Error Allreduce(const Data& send, Data recv, xmpi_op op) override;
  Error Allgather(const Data& send, Data recv) override;
  Error Barrier() override;

 private:
  // pairwiseConnection (network.go:501-506) + its two tagManagers (network.go:448-497)
  struct Peer {
    int dial_fd = -1;    // my data out, the peer's acks in
    int listen_fd = -1;  // the peer's data in, my acks out
    std::mutex mu, write_dial, write_listen;
    std::condition_variable cv;
    std::set<int> send_tags, recv_tags;  // {peer, tag} in use (mpi.go:121-125)
    std::set<int> acked;                 // tags whose ack has arrived
    std::map<int, std::deque<std::vector<uint8_t>>> inbox;  // payloads by tag, awaiting their Receive
    bool acks_closed = false, data_closed = false;  // the reader of that direction has seen the connection end
    std::thread data_reader, ack_reader;
  };
  std::vector<std::unique_ptr<Peer>> peers_;
  int rank_ = 0, size_ = 0;

  std::string accept_peers(int n);
  std::string dial_peers(int n);
  std::string check_peer(const std::string& password, int64_t id, int n) const;
  void reader_loop(int peer, bool acks);
  void close_all();
  Error exchange(const Data& send, std::vector<std::vector<uint8_t>>* all, int only_to);  // only_to < 0: everybody
};

}  // namespace mpi
