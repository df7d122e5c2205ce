// mpi::Network semantics that need no second process: two ranks as two objects in one process (threads), loopback TCP
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

#include "network.hpp"

int main() {
  int bad = 0;
  auto expect = [&](bool c, const char* what) {
    if (!c) {
      printf("FAILED: %s\n", what);
      bad++;
    }
  };
  {
    mpi::Network fresh;
    expect(fresh.Rank() == -1 && fresh.Size() == 0, "Rank / Size before Init");
  }
  const std::vector<std::string> addrs = {":8151", ":8152"};
  {  // wrong password: the handshake refuses on both sides
    mpi::Network a, b;
    a.Addr = addrs[0]; a.Addrs = addrs; a.Password = "x"; a.Timeout = 3; a.NetProto = "tcp";
    b.Addr = addrs[1]; b.Addrs = addrs; b.Password = "y"; b.Timeout = 3; b.NetProto = "tcp";
    mpi::Error ea, eb;
    std::thread ta([&] { ea = a.Init(); }), tb([&] { eb = b.Init(); });
    ta.join();
    tb.join();
    expect((bool)ea && (bool)eb, "mismatched passwords must fail Init");
    expect(a.Size() == 0 && b.Size() == 0, "a failed Init leaves the backend uninitialised");
  }
  {
    const std::vector<std::string> ad2 = {":8153", ":8154"};
    mpi::Network a, b;
    a.Addr = ad2[0]; a.Addrs = ad2; a.Timeout = 10; a.NetProto = "tcp";
    b.Addr = ad2[1]; b.Addrs = ad2; b.Timeout = 10; b.NetProto = "tcp";
    mpi::Error ea, eb;
    std::thread ta([&] { ea = a.Init(); }), tb([&] { eb = b.Init(); });
    ta.join();
    tb.join();
    expect(!ea && !eb && a.Rank() == 0 && b.Rank() == 1 && a.Size() == 2, "Init of two ranks");
    // a Send that is still waiting for its Receive holds {dest, tag}: a second one is refused with TagExists
    std::vector<double> v = {1.5, -2.25, 1e300}, got;
    mpi::Error first, second;
    std::thread s1([&] { first = a.Send(mpi::Slice(v), 1, 5); });
    std::this_thread::sleep_for(std::chrono::milliseconds(200));
    second = a.Send(mpi::Slice(v), 1, 5);
    expect(second.IsTagExists(), "duplicate {dest, tag} -> TagExists");
    // the message waits in the receiver's queue until its Receive is posted (the reference would panic), and the
    // Send returns only then
    mpi::Error r = b.Receive(mpi::Into(&got), 0, 5);
    s1.join();
    expect(!first && !r && got == v, "float64 slice round trip, Receive posted late");
    // receives posted in another order than the sends: nothing deadlocks, nothing is mixed up
    std::vector<int64_t> x1 = {1, 2, 3}, x2 = {-7}, g1, g2;
    std::thread s2([&] { a.Send(mpi::Slice(x1), 1, 1); }), s3([&] { a.Send(mpi::Slice(x2), 1, 2); });
    expect(!b.Receive(mpi::Into(&g2), 0, 2) && !b.Receive(mpi::Into(&g1), 0, 1), "receives in the other order");
    s2.join();
    s3.join();
    expect(g1 == x1 && g2 == x2, "payloads routed by tag");
    // float32 travels widened to float64 and comes back exact; strings and byte slices are different wire types
    std::vector<float> f = {1.0f / 3.0f, -0.0f, 3.4e38f}, fg;
    std::thread s4([&] { a.Send(mpi::Slice(f), 1, 9); });
    expect(!b.Receive(mpi::Into(&fg), 0, 9) && fg == f, "float32 slice exact");
    s4.join();
    std::string str = "hello", sg;
    std::thread s5([&] { b.Send(mpi::Slice(str), 0, 3); });
    expect(!a.Receive(mpi::Into(&sg), 1, 3) && sg == str, "string");
    s5.join();
    std::vector<double> wrong;
    std::thread s6([&] { b.Send(mpi::Slice(str), 0, 4); });
    expect((bool)a.Receive(mpi::Into(&wrong), 1, 4), "receiving a string into a []float64 is an error, as gob's would be");
    s6.join();
    a.Finalize();
    b.Finalize();
    expect(a.Rank() == -1 && b.Size() == 0, "Finalize");
  }
  printf(bad ? "FAILED\n" : "ok\n");
  return bad ? 1 : 0;
}
