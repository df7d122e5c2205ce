// refpath.cpp -- CPU restatement of the reference's hot path: mpi.Network over loopback TCP with
// gob framing.  TEST INFRASTRUCTURE / CPU BASELINE ONLY; nothing under mpi_amd/ links this.
//
// The reference binary cannot be built here (no Go toolchain in the image), so bench.py's
// cpu_baseline leg ("kind": "port") times this program instead, on the GPU box's host cores.
// It follows the reference function by function:
//   Init            network.go:53-65    flags -> sort addresses -> rank = index (network.go:94-109)
//   startConnections network.go:122-159 listen + dial every peer concurrently, 2 conns per pair
//   listenHandshake  network.go:211-263 accept, decode initialMessage, check password/id, reply
//   dialHandshake    network.go:297-339 retry every 100 ms until the peer listens
//   Send             network.go:518-572 gob(data) -> message{Tag,Bytes} on the dial conn -> wait ack
//   Receive          network.go:575-602 receiveReader decodes one message, routes by tag, acks
//   local            network.go:388-446 self-send through an in-process rendezvous
//   tagManager       network.go:448-497
// Collectives do not exist upstream (mpi.go:130); "allreduce" here is what a reference user
// writes: the all-to-all exchange of examples/helloworld/helloworld.go:53-81 (one goroutine per
// Send and per Receive, self included) followed by a host sum in rank order.
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "gob_codec.h"

namespace ref {

typedef std::vector<uint8_t> Bytes;

// unbuffered Go channel of []byte: the sender blocks until a receiver took the value
class Chan {
 public:
  void send(Bytes b) {
    std::unique_lock<std::mutex> l(mu_);
    cv_.wait(l, [&] { return !full_; });
    val_ = std::move(b);
    full_ = true;
    cv_.notify_all();
    cv_.wait(l, [&] { return taken_; });
    taken_ = false;
    full_ = false;
    cv_.notify_all();
  }
  Bytes recv() {
    std::unique_lock<std::mutex> l(mu_);
    cv_.wait(l, [&] { return full_ && !taken_; });
    Bytes b = std::move(val_);
    taken_ = true;
    cv_.notify_all();
    return b;
  }

 private:
  std::mutex mu_;
  std::condition_variable cv_;
  Bytes val_;
  bool full_ = false, taken_ = false;
};

// network.go:448-497
class TagManager {
 public:
  void Register(int tag) {
    std::lock_guard<std::mutex> g(mu_);
    if (m_.count(tag)) throw std::runtime_error("tag " + std::to_string(tag) + " already exists on the connection");
    m_[tag] = std::make_shared<Chan>();
  }
  void Delete(int tag) {
    std::lock_guard<std::mutex> g(mu_);
    if (!m_.count(tag)) throw std::runtime_error("attempt to delete non-existant key");
    m_.erase(tag);
  }
  std::shared_ptr<Chan> Channel(int tag) {
    std::lock_guard<std::mutex> g(mu_);
    auto it = m_.find(tag);
    if (it == m_.end()) throw std::runtime_error("attempt to return chan from non-existant tag");
    return it->second;
  }

 private:
  std::mutex mu_;
  std::map<int, std::shared_ptr<Chan>> m_;
};

// network.go:388-446: whichever of Send/Receive arrives first creates the channel
class Local {
 public:
  void Send(int tag, Bytes b) {
    std::shared_ptr<Chan> c;
    {
      std::lock_guard<std::mutex> g(mu_);
      auto it = m_.find(tag);
      if (it == m_.end()) it = m_.emplace(tag, std::make_shared<Chan>()).first;
      c = it->second;
    }
    c->send(std::move(b));
    std::lock_guard<std::mutex> g(mu_);
    m_.erase(tag);
  }
  Bytes Receive(int tag) {
    std::shared_ptr<Chan> c;
    {
      std::lock_guard<std::mutex> g(mu_);
      auto it = m_.find(tag);
      if (it == m_.end()) it = m_.emplace(tag, std::make_shared<Chan>()).first;
      c = it->second;
    }
    return c->recv();
  }

 private:
  std::mutex mu_;
  std::map<int, std::shared_ptr<Chan>> m_;
};

static void write_all(int fd, const uint8_t* p, size_t n) {
  while (n) {
    ssize_t w = ::send(fd, p, n, MSG_NOSIGNAL);
    if (w <= 0) throw std::runtime_error("socket write failed");
    p += w;
    n -= (size_t)w;
  }
}

static void read_all(int fd, uint8_t* p, size_t n) {
  while (n) {
    ssize_t r = ::recv(fd, p, n, 0);
    if (r <= 0) throw std::runtime_error("socket read failed");
    p += r;
    n -= (size_t)r;
  }
}

// one framed gob message (length prefix + body) appended to out
static void read_frame(int fd, Bytes* out) {
  uint8_t c;
  read_all(fd, &c, 1);
  out->push_back(c);
  uint64_t len = c;
  if (c >= 128) {
    int nb = 256 - (int)c;
    uint8_t tmp[8];
    read_all(fd, tmp, (size_t)nb);
    len = 0;
    for (int i = 0; i < nb; i++) {
      len = (len << 8) | tmp[i];
      out->push_back(tmp[i]);
    }
  }
  size_t at = out->size();
  out->resize(at + (size_t)len);
  read_all(fd, out->data() + at, (size_t)len);
}

// what a fresh gob.Decoder(conn).Decode(&v) consumes: type definitions, then one value message
static void read_value(int fd, Bytes* out) {
  for (;;) {
    size_t start = out->size();
    read_frame(fd, out);
    gob::Reader r(out->data() + start, out->size() - start), body(nullptr, 0);
    gob::next_message(r, &body);
    if (body.int_() >= 0) return;
  }
}

struct Pair {
  int dial = -1, listen = -1;
  TagManager receivetags, sendtags;
};

class Network {
 public:
  std::string Addr, Password;
  std::vector<std::string> Addrs;
  double TimeoutS = 0;

  int Rank() const { return n_ == 0 ? -1 : rank_; }
  int Size() const { return n_; }

  void Init() {
    if (Addrs.empty()) {
      Addr = ":5000";
      Addrs = {":5000"};
    }
    std::sort(Addrs.begin(), Addrs.end());  // lexicographic, as sort.Strings (network.go:95)
    for (size_t i = 0; i + 1 < Addrs.size(); i++)
      if (Addrs[i] == Addrs[i + 1]) throw std::runtime_error("network addresses not unique");
    auto it = std::lower_bound(Addrs.begin(), Addrs.end(), Addr);
    if (it == Addrs.end() || *it != Addr) throw std::runtime_error("mpi init: local ip address not in global list");
    rank_ = (int)(it - Addrs.begin());
    n_ = (int)Addrs.size();
    conns_.resize((size_t)n_);
    for (auto& p : conns_) p.reset(new Pair);
    std::string lerr, derr;
    std::thread tl([&] { try { listenAll(); } catch (std::exception& e) { lerr = e.what(); } });
    std::thread td([&] { try { dialAll(); } catch (std::exception& e) { derr = e.what(); } });
    tl.join();
    td.join();
    if (!lerr.empty()) throw std::runtime_error("error listening: " + lerr);
    if (!derr.empty()) throw std::runtime_error(derr);
  }

  void Finalize() {
    for (auto& p : conns_) {
      if (p->dial >= 0) close(p->dial);
      if (p->listen >= 0) close(p->listen);
    }
    if (lfd_ >= 0) close(lfd_);
  }

  // `encoded` = gob.NewEncoder(&buf).Encode(data) already done by the caller (network.go:537-542)
  void Send(const Bytes& encoded, int dest, int tag) {
    TagManager& mgr = conns_[(size_t)dest]->sendtags;
    mgr.Register(tag);
    if (dest == rank_) {  // network.go:545-548 (the reference leaks the tag here: quirk Q1, not replicated)
      local_.Send(tag, encoded);
      mgr.Delete(tag);
      return;
    }
    Pair* p = conns_[(size_t)dest].get();
    std::thread ack([p, &mgr] {  // network.go:551-559
      Bytes raw;
      read_value(p->dial, &raw);
      int64_t t;
      Bytes payload;
      if (!gob::decode_message(raw.data(), raw.size(), &t, &payload, nullptr)) throw std::runtime_error("bad ack");
      mgr.Channel((int)t)->send(std::move(payload));
    });
    gob::Buf wire;
    gob::encode_message(wire, tag, encoded.data(), encoded.size());  // network.go:562-563
    write_all(p->dial, wire.data(), wire.size());
    mgr.Channel(tag)->recv();  // network.go:569
    ack.join();
    mgr.Delete(tag);
  }

  Bytes Receive(int source, int tag) {
    if (source == rank_) return local_.Receive(tag);
    TagManager& mgr = conns_[(size_t)source]->receivetags;
    mgr.Register(tag);
    Pair* p = conns_[(size_t)source].get();
    std::thread reader([p, &mgr] {  // receiveReader, network.go:607-625
      Bytes raw;
      read_value(p->listen, &raw);
      int64_t t;
      Bytes payload;
      if (!gob::decode_message(raw.data(), raw.size(), &t, &payload, nullptr)) throw std::runtime_error("bad message");
      mgr.Channel((int)t)->send(std::move(payload));
      gob::Buf reply;
      gob::encode_message(reply, t, nullptr, 0);
      write_all(p->listen, reply.data(), reply.size());
    });
    Bytes b = mgr.Channel(tag)->recv();
    reader.join();
    mgr.Delete(tag);
    return b;
  }

 private:
  int rank_ = 0, n_ = 0, lfd_ = -1;
  std::vector<std::unique_ptr<Pair>> conns_;
  Local local_;

  static int port_of(const std::string& a) { return atoi(a.substr(a.rfind(':') + 1).c_str()); }

  int checkPasswordAndId(const std::string& pw, int64_t id) {  // network.go:343-351
    if (pw != Password) throw std::runtime_error("bad password");
    if (id >= n_ || id < 0 || id == rank_) throw std::runtime_error("bad id: " + std::to_string(id));
    return (int)id;
  }

  void listenAll() {  // network.go:163-263
    lfd_ = socket(AF_INET, SOCK_STREAM, 0);
    int one = 1;
    setsockopt(lfd_, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    sockaddr_in sa{};
    sa.sin_family = AF_INET;
    sa.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    sa.sin_port = htons((uint16_t)port_of(Addr));
    if (bind(lfd_, (sockaddr*)&sa, sizeof sa) != 0 || listen(lfd_, 64) != 0) throw std::runtime_error("listen failed");
    std::vector<std::thread> ts;
    std::mutex emu;
    std::string err;
    for (int i = 0; i < n_ - 1; i++)
      ts.emplace_back([&] {
        try {
          int fd = accept(lfd_, nullptr, nullptr);
          if (fd < 0) throw std::runtime_error("error accepting");
          int one2 = 1;
          setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one2, sizeof one2);
          Bytes raw;
          read_value(fd, &raw);
          std::string pw;
          int64_t id;
          if (!gob::decode_initial(raw.data(), raw.size(), &pw, &id, nullptr)) throw std::runtime_error("bad handshake");
          int peer = checkPasswordAndId(pw, id);
          conns_[(size_t)peer]->listen = fd;
          gob::Buf reply;
          gob::encode_initial(reply, Password, rank_);
          write_all(fd, reply.data(), reply.size());
        } catch (std::exception& e) {
          std::lock_guard<std::mutex> g(emu);
          err = e.what();
        }
      });
    for (auto& t : ts) t.join();
    if (!err.empty()) throw std::runtime_error(err);
  }

  void dialAll() {  // network.go:265-339
    std::vector<std::thread> ts;
    std::mutex emu;
    std::string err;
    for (int i = 0; i < n_; i++) {
      if (i == rank_) continue;
      ts.emplace_back([&, i] {
        try {
          auto t0 = std::chrono::steady_clock::now();
          int fd = -1;
          for (;;) {  // 100 ms ticker (network.go:298-312)
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
            fd = socket(AF_INET, SOCK_STREAM, 0);
            sockaddr_in sa{};
            sa.sin_family = AF_INET;
            sa.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
            sa.sin_port = htons((uint16_t)port_of(Addrs[(size_t)i]));
            if (connect(fd, (sockaddr*)&sa, sizeof sa) == 0) break;
            close(fd);
            double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (TimeoutS > 0 && el > TimeoutS) throw std::runtime_error("dial timed out");
          }
          int one = 1;
          setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
          gob::Buf hello;
          gob::encode_initial(hello, Password, rank_);
          write_all(fd, hello.data(), hello.size());
          Bytes raw;
          read_value(fd, &raw);
          std::string pw;
          int64_t id;
          if (!gob::decode_initial(raw.data(), raw.size(), &pw, &id, nullptr)) throw std::runtime_error("bad handshake");
          int peer = checkPasswordAndId(pw, id);
          conns_[(size_t)peer]->dial = fd;
        } catch (std::exception& e) {
          std::lock_guard<std::mutex> g(emu);
          err = e.what();
        }
      });
    }
    for (auto& t : ts) t.join();
    if (!err.empty()) throw std::runtime_error(err);
  }
};

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// every rank sends its slice to every rank (self included) and receives everybody's, concurrently
// (helloworld.go:53-81), then folds in rank order: what a reference user has instead of Allreduce
static void user_allreduce_f32(Network& net, const std::vector<float>& mine, std::vector<float>* out) {
  const int n = net.Size();
  std::vector<std::vector<float>> got((size_t)n);
  std::vector<std::thread> ts;
  for (int i = 0; i < n; i++)
    ts.emplace_back([&, i] {
      gob::Buf enc;
      gob::encode_f32_slice(enc, mine.data(), mine.size());
      net.Send(enc, i, 0);
    });
  for (int i = 0; i < n; i++)
    ts.emplace_back([&, i] {
      Bytes b = net.Receive(i, 0);
      if (!gob::decode_f32_slice(b.data(), b.size(), &got[(size_t)i])) throw std::runtime_error("decode failed");
    });
  for (auto& t : ts) t.join();
  *out = got[0];
  for (int r = 1; r < n; r++)
    for (size_t k = 0; k < out->size(); k++) (*out)[k] = (*out)[k] + got[(size_t)r][k];
}

}  // namespace ref

// ---- C hooks for tests/test_gob.py (codec known answers) -------------------------------------
extern "C" {

static size_t emit(const gob::Buf& b, uint8_t* out, size_t cap) {
  if (out && cap >= b.size()) memcpy(out, b.data(), b.size());
  return b.size();
}

size_t gobx_uint(uint64_t v, uint8_t* out, size_t cap) { gob::Buf b; gob::put_uint(b, v); return emit(b, out, cap); }
size_t gobx_int(int64_t v, uint8_t* out, size_t cap) { gob::Buf b; gob::put_int(b, v); return emit(b, out, cap); }
size_t gobx_float(double v, uint8_t* out, size_t cap) { gob::Buf b; gob::put_float(b, v); return emit(b, out, cap); }

// type Point struct{X, Y int}; the document's worked example
size_t gobx_point(int64_t x, int64_t y, uint8_t* out, size_t cap) {
  gob::Buf o, b;
  gob::def_struct(o, 65, "Point", {{"X", gob::tInt}, {"Y", gob::tInt}});
  gob::put_int(b, 65);
  int last = -1;
  if (x) { b.push_back((uint8_t)(0 - last)); gob::put_int(b, x); last = 0; }
  if (y) { b.push_back((uint8_t)(1 - last)); gob::put_int(b, y); }
  b.push_back(0);
  gob::frame(o, b);
  return emit(o, out, cap);
}

size_t gobx_encode_f64(const double* v, size_t n, uint8_t* out, size_t cap) { gob::Buf b; gob::encode_f64_slice(b, v, n); return emit(b, out, cap); }
size_t gobx_encode_f32(const float* v, size_t n, uint8_t* out, size_t cap) { gob::Buf b; gob::encode_f32_slice(b, v, n); return emit(b, out, cap); }
size_t gobx_encode_i64(const int64_t* v, size_t n, uint8_t* out, size_t cap) { gob::Buf b; gob::encode_i64_slice(b, v, n); return emit(b, out, cap); }
size_t gobx_encode_bytes(const uint8_t* v, size_t n, uint8_t* out, size_t cap) { gob::Buf b; gob::encode_bytes(b, v, n); return emit(b, out, cap); }
size_t gobx_encode_message(int64_t tag, const uint8_t* v, size_t n, uint8_t* out, size_t cap) { gob::Buf b; gob::encode_message(b, tag, v, n); return emit(b, out, cap); }

long gobx_decode_f64(const uint8_t* p, size_t n, double* out, size_t cap) {
  std::vector<double> v;
  if (!gob::decode_f64_slice(p, n, &v) || v.size() > cap) return -1;
  memcpy(out, v.data(), v.size() * 8);
  return (long)v.size();
}
long gobx_decode_f32(const uint8_t* p, size_t n, float* out, size_t cap) {
  std::vector<float> v;
  if (!gob::decode_f32_slice(p, n, &v) || v.size() > cap) return -1;
  memcpy(out, v.data(), v.size() * 4);
  return (long)v.size();
}
long gobx_decode_i64(const uint8_t* p, size_t n, int64_t* out, size_t cap) {
  std::vector<int64_t> v;
  if (!gob::decode_i64_slice(p, n, &v) || v.size() > cap) return -1;
  memcpy(out, v.data(), v.size() * 8);
  return (long)v.size();
}
long gobx_decode_bytes(const uint8_t* p, size_t n, uint8_t* out, size_t cap) {
  std::vector<uint8_t> v;
  if (!gob::decode_bytes(p, n, &v) || v.size() > cap) return -1;
  memcpy(out, v.data(), v.size());
  return (long)v.size();
}
long gobx_decode_message(const uint8_t* p, size_t n, int64_t* tag, uint8_t* out, size_t cap) {
  std::vector<uint8_t> v;
  if (!gob::decode_message(p, n, tag, &v, nullptr) || v.size() > cap) return -1;
  if (!v.empty()) memcpy(out, v.data(), v.size());
  return (long)v.size();
}

}  // extern "C"

#ifdef REFPATH_MAIN
// refpath_bin <mode> -mpi-addr :P -mpi-alladdr :P0,:P1,... [count] [reps]
//   mode = helloworld | bounce | allreduce_f32
int main(int argc, char** argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: %s <mode> -mpi-addr A -mpi-alladdr CSV [count] [reps]\n", argv[0]);
    return 2;
  }
  std::string mode = argv[1];
  ref::Network net;
  std::vector<std::string> rest;
  for (int i = 2; i < argc; i++) {
    std::string a = argv[i];
    if (a == "-mpi-addr" && i + 1 < argc) net.Addr = argv[++i];
    else if (a == "-mpi-alladdr" && i + 1 < argc) {
      std::string csv = argv[++i];
      size_t p = 0;
      while (p <= csv.size()) {
        size_t q = csv.find(',', p);
        if (q == std::string::npos) q = csv.size();
        net.Addrs.push_back(csv.substr(p, q - p));
        p = q + 1;
      }
    } else if (a == "-mpi-password" && i + 1 < argc) net.Password = argv[++i];
    else rest.push_back(a);
  }
  size_t count = rest.size() > 0 ? (size_t)atoll(rest[0].c_str()) : 1024;
  int reps = rest.size() > 1 ? atoi(rest[1].c_str()) : 3;
  try {
    net.TimeoutS = 30;
    net.Init();
    const int rank = net.Rank(), size = net.Size();
    if (mode == "helloworld") {  // examples/helloworld/helloworld.go
      printf("Hello world, I'm node %d in a land with %d nodes\n", rank, size);
      std::vector<std::thread> ts;
      std::mutex pm;
      for (int i = 0; i < size; i++)
        ts.emplace_back([&, i] {
          char s[128];
          if (i == rank) snprintf(s, sizeof s, "\"I'm just node %d talking to myself\"", rank);
          else snprintf(s, sizeof s, "\"Hello node %d, I'm node %d\"", i, rank);
          gob::Buf enc;
          gob::encode_string(enc, s);
          net.Send(enc, i, 0);
        });
      for (int i = 0; i < size; i++)
        ts.emplace_back([&, i] {
          ref::Bytes b = net.Receive(i, 0);
          std::string s;
          if (!gob::decode_string(b.data(), b.size(), &s)) throw std::runtime_error("decode");
          std::lock_guard<std::mutex> g(pm);
          printf("I, node %d, received a message: %s\n", rank, s.c_str());
        });
      for (auto& t : ts) t.join();
    } else if (mode == "bounce") {  // examples/bounce/bounce.go:83-151
      const size_t lens[] = {0, 1, 10, 100, 1000, 10000, 100000, 1000000, 10000000};
      const bool even = rank % 2 == 0;
      std::vector<uint8_t> msg(10000000);
      for (size_t i = 0; i < msg.size(); i++) msg[i] = (uint8_t)(i * 2654435761u >> 24);
      std::vector<double> msgf(10000000 / 8);
      for (size_t i = 0; i < msgf.size(); i++) msgf[i] = (double)((i * 2654435761u) & 0xFFFFFF) / 16777216.0;
      printf("{\"mode\":\"bounce\",\"rank\":%d", rank);
      std::string tb = "\"bytes_us\":[", tf = "\"float64_us\":[";
      for (size_t li = 0; li < sizeof lens / sizeof lens[0] && lens[li] <= count; li++) {
        const size_t l = lens[li];
        double accb = 0, accf = 0;
        for (int j = 0; j < reps; j++) {
          double t0 = ref::now_s();
          std::vector<uint8_t> rcv;
          if (even) {
            gob::Buf e; gob::encode_bytes(e, msg.data(), l); net.Send(e, rank + 1, 0);
            ref::Bytes b = net.Receive(rank + 1, 0); gob::decode_bytes(b.data(), b.size(), &rcv);
            if (rcv.size() != l || memcmp(rcv.data(), msg.data(), l)) throw std::runtime_error("message not the same");
          } else {
            ref::Bytes b = net.Receive(rank - 1, 0); gob::decode_bytes(b.data(), b.size(), &rcv);
            gob::Buf e; gob::encode_bytes(e, rcv.data(), rcv.size()); net.Send(e, rank - 1, 0);
          }
          accb += ref::now_s() - t0;
          t0 = ref::now_s();
          std::vector<double> rf;
          if (even) {
            gob::Buf e; gob::encode_f64_slice(e, msgf.data(), l / 8); net.Send(e, rank + 1, 0);
            ref::Bytes b = net.Receive(rank + 1, 0); gob::decode_f64_slice(b.data(), b.size(), &rf);
            if (rf.size() != l / 8 || memcmp(rf.data(), msgf.data(), l / 8 * 8)) throw std::runtime_error("message not the same");
          } else {
            ref::Bytes b = net.Receive(rank - 1, 0); gob::decode_f64_slice(b.data(), b.size(), &rf);
            gob::Buf e; gob::encode_f64_slice(e, rf.data(), rf.size()); net.Send(e, rank - 1, 0);
          }
          accf += ref::now_s() - t0;
        }
        char tmp[64];
        snprintf(tmp, sizeof tmp, "%s%.1f", li ? "," : "", 1e6 * accb / reps); tb += tmp;
        snprintf(tmp, sizeof tmp, "%s%.1f", li ? "," : "", 1e6 * accf / reps); tf += tmp;
      }
      printf(",%s],%s]}\n", tb.c_str(), tf.c_str());
    } else if (mode == "allreduce_f32") {
      std::vector<float> mine(count), out;
      for (size_t i = 0; i < count; i++) mine[i] = (float)((i * 7 + (size_t)rank * 13) % 1000) / 1024.0f;
      double best = 1e30, tot = 0;
      for (int j = 0; j < reps; j++) {
        double t0 = ref::now_s();
        ref::user_allreduce_f32(net, mine, &out);
        double dt = ref::now_s() - t0;
        best = std::min(best, dt);
        tot += dt;
      }
      // known answer: sum over ranks of ((7i + 13r) mod 1000)/1024, in rank order in float32
      size_t bad = 0;
      for (size_t i = 0; i < count; i += 997) {
        float acc = (float)((i * 7) % 1000) / 1024.0f;
        for (int r = 1; r < size; r++) acc = acc + (float)((i * 7 + (size_t)r * 13) % 1000) / 1024.0f;
        bad += (acc != out[i]);
      }
      printf("{\"mode\":\"allreduce_f32\",\"rank\":%d,\"ranks\":%d,\"count\":%zu,\"reps\":%d,\"mean_s\":%.6f,\"best_s\":%.6f,\"bad\":%zu}\n",
             rank, size, count, reps, tot / reps, best, bad);
      if (bad) return 1;
    } else {
      fprintf(stderr, "unknown mode %s\n", mode.c_str());
      return 2;
    }
    net.Finalize();
  } catch (std::exception& e) {
    fprintf(stderr, "rank %d: %s\n", net.Rank(), e.what());
    return 1;
  }
  return 0;
}
#endif
