// network.cpp -- mpi::Network: the reference's own backend (type Network, network.go:25-625) in C++, speaking the
// reference's wire protocol -- gob-framed initialMessage handshake, message{Tag, Bytes} + ack per Send -- over TCP.
// A rank running this backend interoperates with ranks running the reference's (Go) backend, which is what makes
// it SURVEY.md section 8 (f) row 2: mixed CPU / GPU jobs, multi-node, and an end-to-end check of this project's
// understanding of the reference's Send / Receive against something that speaks its format.  Host memory only
// (it is the CPU path); the xGMI backend is mpi::XGMI.  Nothing here touches oracle/.
//
//   reference                                        here
//   Init -> useFlags, assignRanks, startConnections  Network::Init                 network.go:53-159
//   listenHandshake / dialHandshake                  accept_peers / dial_peers     network.go:211-339
//   passwordAndId                                    check_peer                    network.go:343-351
//   Send: encode, write message, wait for ack        Network::Send                 network.go:518-572
//   Receive + receiveReader: read, route, ack        reader_loop + Network::Receive network.go:575-625
//   local (self-send hand-off)                       the same queues, no socket    network.go:388-446
// Deliberate differences: a duplicate {peer, tag} is an error value
// (mpi.TagExists, declared at mpi.go:172-182) instead of a panic; a message whose Receive has not been posted
// yet waits in a queue (the reference panics, network.go:614 -> :493) and is acknowledged when it is taken, so
// Send still returns only after the matching Receive consumed it; one reader thread per connection decodes
// messages in order (the reference starts a reader per call and lets them race on the conn).
#include "network.hpp"

#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstring>
#include <new>

#include "gobwire.hpp"

namespace mpi {

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

bool write_all(int fd, const uint8_t* p, size_t n) {
  while (n) {
    const ssize_t w = ::send(fd, p, n, MSG_NOSIGNAL);
    if (w < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    p += w;
    n -= (size_t)w;
  }
  return true;
}

bool read_exact(int fd, uint8_t* p, size_t n) {
  while (n) {
    const ssize_t r = ::recv(fd, p, n, 0);
    if (r == 0) return false;
    if (r < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    p += r;
    n -= (size_t)r;
  }
  return true;
}

// the largest message body accepted: a handshake is a few dozen bytes (4 KiB is generous: a stranger's first bytes are
// read BEFORE its password is checked and must not be able to make this rank allocate at will); a data message is bounded
// by what a rank can hold (64 GiB)
constexpr uint64_t kMaxHandshakeBytes = 4096, kMaxMessageBytes = (uint64_t)64 << 30;

// one gob message off the wire: the length prefix, then the body; appended to `stream` as it arrived
bool read_gob_message(int fd, gobwire::Bytes* stream, bool* is_value, uint64_t max_len) {
  uint8_t first;
  if (!read_exact(fd, &first, 1)) return false;
  stream->push_back(first);
  uint64_t len = first;
  if (first >= 0x80) {
    const int n = 0x100 - (int)first;
    if (n > 8) return false;
    uint8_t be[8];
    if (!read_exact(fd, be, (size_t)n)) return false;
    stream->insert(stream->end(), be, be + n);
    len = 0;
    for (int k = 0; k < n; k++) len = (len << 8) | be[k];
  }
  if (len > max_len) return false;
  const size_t at = stream->size();
  try {
    stream->resize(at + (size_t)len);
  } catch (const std::bad_alloc&) {  // (the accept / dial threads must never terminate the process)
    return false;
  }
  if (!read_exact(fd, stream->data() + at, (size_t)len)) return false;
  gobwire::Reader body(stream->data() + at, (size_t)len);
  const int64_t id = body.i();
  *is_value = body.ok() && id > 0;
  return body.ok();
}

// everything a fresh gob.Encoder sent for ONE Encode call: type definitions, then the value
bool read_gob_value(int fd, gobwire::Bytes* stream, uint64_t max_len = kMaxMessageBytes) {
  stream->clear();
  for (int messages = 0; messages < 64; messages++) {  // (a handful of type definitions precede a value, never dozens)
    bool is_value = false;
    if (!read_gob_message(fd, stream, &is_value, max_len)) return false;
    if (is_value) return true;
  }
  return false;
}

// "host:port" / ":port" as the reference's net.Listen / net.Dial take them
bool split_addr(const std::string& a, std::string* host, std::string* port) {
  const size_t c = a.rfind(':');
  if (c == std::string::npos) return false;
  *host = a.substr(0, c);
  *port = a.substr(c + 1);
  return !port->empty();
}

}  // namespace

Network::~Network() { Finalize(); }

int Network::Rank() { return size_ == 0 ? -1 : rank_; }  // network.go:41-46
int Network::Size() { return size_; }                    // network.go:48-50

Error Network::Init() {
  // useFlags (network.go:69-90)
  if (Password.empty()) Password = FlagPassword;
  if (Timeout == 0) Timeout = FlagInitTimeout;
  if (Addr.empty()) Addr = FlagAddr;
  if (Addrs.empty()) Addrs = FlagAllAddrs;
  if (NetProto.empty()) NetProto = FlagProtocol;
  if (NetProto != "tcp" && NetProto != "tcp4") return Error(XMPI_ERR_UNSUPPORTED, "mpi init: protocol " + NetProto + " is not supported (tcp)");
  if (Addrs.empty()) {  // network.go:55-58
    Addr = ":5000";
    Addrs = {":5000"};
  }
  // assignRanks (network.go:94-109)
  std::sort(Addrs.begin(), Addrs.end());
  for (size_t i = 0; i + 1 < Addrs.size(); i++)
    if (Addrs[i] == Addrs[i + 1]) return Error(XMPI_ERR_ARG, "network addresses not unique");
  auto it = std::lower_bound(Addrs.begin(), Addrs.end(), Addr);
  if (it == Addrs.end() || *it != Addr)
    return Error(XMPI_ERR_ARG, "mpi init: local ip address not in global list. Local address is: " + Addr);
  rank_ = (int)(it - Addrs.begin());
  const int n = (int)Addrs.size();
  peers_.clear();
  for (int i = 0; i < n; i++) peers_.emplace_back(new Peer);
  // startConnections (network.go:122-159): listen and dial concurrently
  std::string lerr, derr;
  std::thread tl([&] { lerr = accept_peers(n); });
  std::thread td([&] { derr = dial_peers(n); });
  tl.join();
  td.join();
  if (!lerr.empty() || !derr.empty()) {
    close_all();
    return Error(XMPI_ERR_BOOTSTRAP, lerr.empty() ? derr : lerr);
  }
  size_ = n;
  for (int i = 0; i < n; i++) {
    if (i == rank_) continue;
    peers_[(size_t)i]->data_reader = std::thread([this, i] { reader_loop(i, /*acks=*/false); });
    peers_[(size_t)i]->ack_reader = std::thread([this, i] { reader_loop(i, /*acks=*/true); });
  }
  return Error();
}

// listener side of the handshake (network.go:163-263)
std::string Network::accept_peers(int n) {
  if (n == 1) return "";
  std::string host, port;
  if (!split_addr(Addr, &host, &port)) return "error listening: bad address " + Addr;
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_INET;
  hints.ai_socktype = SOCK_STREAM;
  hints.ai_flags = AI_PASSIVE;
  if (getaddrinfo(host.empty() ? nullptr : host.c_str(), port.c_str(), &hints, &res) != 0 || !res)
    return "error listening: cannot resolve " + Addr;
  const int ls = ::socket(res->ai_family, res->ai_socktype, 0);
  int one = 1;
  setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
  if (ls < 0 || ::bind(ls, res->ai_addr, res->ai_addrlen) != 0 || ::listen(ls, n + 16) != 0) {
    freeaddrinfo(res);
    if (ls >= 0) ::close(ls);
    return std::string("error listening: ") + strerror(errno);
  }
  freeaddrinfo(res);
  const double t0 = now_s();
  int have = 0;
  std::string err;
  while (have < n - 1 && err.empty()) {
    if (Timeout > 0) {  // network.go:223-234
      const double left = Timeout - (now_s() - t0);
      if (left <= 0) {
        err = "listen timed out";
        break;
      }
      timeval tv{(time_t)left, (suseconds_t)((left - (double)(time_t)left) * 1e6)};
      setsockopt(ls, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
    }
    const int fd = ::accept(ls, nullptr, nullptr);
    if (fd < 0) {
      if (errno == EINTR) continue;
      err = (errno == EAGAIN || errno == EWOULDBLOCK) ? "listen timed out" : std::string("error accepting: ") + strerror(errno);
      break;
    }
    setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    {  // A real peer says hello the moment it is connected (dial_peers below; network.go:318).  The accept loop is serial, so
       // a connection that says nothing holds up every peer in the backlog: it gets two seconds, not the job's timeout.
      const double left = std::min(2.0, Timeout > 0 ? std::max(0.05, Timeout - (now_s() - t0)) : 2.0);
      timeval tv{(time_t)left, (suseconds_t)((left - (double)(time_t)left) * 1e6)};
      setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
    }
    gobwire::Bytes in;
    std::string pw;
    int64_t id = -1;
    if (!read_gob_value(fd, &in, kMaxHandshakeBytes) || !gobwire::parse_initial(in.data(), in.size(), &pw, &id)) {
      // a stray connection (a port scanner, a rank of another job): not this job's problem -- keep listening.  The reference
      // fails Init here ("error decoding initial message", network.go:242-246); the deviation is logged, not silent.
      fprintf(stderr, "mpi: rank %d dropped a connection on %s whose first message was not a handshake\n", rank_, Addr.c_str());
      ::close(fd);
      continue;
    }
    const std::string bad = check_peer(pw, id, n);
    if (!bad.empty()) {
      ::close(fd);
      err = bad;
      break;
    }
    if (peers_[(size_t)id]->listen_fd >= 0) {  // that rank has shaken hands already: a duplicate must not replace it
      ::close(fd);
      err = "two peers claim to be rank " + std::to_string(id);
      break;
    }
    {
      timeval none{0, 0};  // the data path blocks for as long as it takes (network.go: no deadlines after Init)
      setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &none, sizeof none);
    }
    peers_[(size_t)id]->listen_fd = fd;  // the peer's data in, my acks out (network.go:255)
    const gobwire::Bytes reply = gobwire::initial_message(Password, rank_);
    if (!write_all(fd, reply.data(), reply.size())) err = "error encoding initial reply";
    have++;
  }
  ::close(ls);
  return err;
}

// dialer side (network.go:265-339): retry every 100 ms until the peer listens or the timeout expires
std::string Network::dial_peers(int n) {
  const double t0 = now_s();
  for (int p = 0; p < n; p++) {
    if (p == rank_) continue;
    std::string host, port;
    if (!split_addr(Addrs[(size_t)p], &host, &port)) return "bad address " + Addrs[(size_t)p];
    if (host.empty()) host = "127.0.0.1";
    int fd = -1;
    for (;;) {
      addrinfo hints{}, *res = nullptr;
      hints.ai_family = AF_INET;
      hints.ai_socktype = SOCK_STREAM;
      if (getaddrinfo(host.c_str(), port.c_str(), &hints, &res) == 0 && res) {
        fd = ::socket(res->ai_family, res->ai_socktype, 0);
        if (fd >= 0 && ::connect(fd, res->ai_addr, res->ai_addrlen) != 0) {
          ::close(fd);
          fd = -1;
        }
        freeaddrinfo(res);
      }
      if (fd >= 0) break;
      if (Timeout > 0 && now_s() - t0 > Timeout) return "dial timed out: " + Addrs[(size_t)p];
      if (Timeout <= 0 && now_s() - t0 > 3600) return "dial gave up: " + Addrs[(size_t)p];
      usleep(100000);  // network.go:298
    }
    int one = 1;
    setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    const gobwire::Bytes hello = gobwire::initial_message(Password, rank_);
    gobwire::Bytes in;
    std::string pw;
    int64_t id = -1;
    if (!write_all(fd, hello.data(), hello.size()) || !read_gob_value(fd, &in, kMaxHandshakeBytes) ||
        !gobwire::parse_initial(in.data(), in.size(), &pw, &id)) {
      ::close(fd);
      return "error in the dial handshake with " + Addrs[(size_t)p];
    }
    const std::string bad = check_peer(pw, id, n);
    if (!bad.empty() || id != p) {
      ::close(fd);
      return bad.empty() ? "peer answered with an unexpected id" : bad;
    }
    peers_[(size_t)p]->dial_fd = fd;  // my data out, the peer's acks in (network.go:337)
  }
  return "";
}

std::string Network::check_peer(const std::string& password, int64_t id, int n) const {  // network.go:343-351
  if (password != Password) return "bad password";
  if (id < 0 || id >= n || id == rank_) return "bad id";
  return "";
}

// One thread per connection and direction.  acks == false: the peer's data messages on listen_fd -- queue the
// payload for the Receive with that tag.  acks == true: the peer's acknowledgements on dial_fd -- release the
// Send with that tag (network.go:551-559).
void Network::reader_loop(int peer, bool acks) {
  Peer& P = *peers_[(size_t)peer];
  const int fd = acks ? P.dial_fd : P.listen_fd;
  gobwire::Bytes stream, payload;
  for (;;) {
    int64_t tag = 0;
    if (!read_gob_value(fd, &stream) || !gobwire::parse_tagged(stream.data(), stream.size(), &tag, &payload)) break;
    std::lock_guard<std::mutex> g(P.mu);
    if (acks) P.acked.insert((int)tag);
    else P.inbox[(int)tag].push_back(std::move(payload));
    payload.clear();
    P.cv.notify_all();
  }
  std::lock_guard<std::mutex> g(P.mu);
  (acks ? P.acks_closed : P.data_closed) = true;  // per direction: a peer that has finished closes both connections,
  P.cv.notify_all();                              // but the acks it sent before are still to be read from the other one
}

Error Network::Send(const Data& d, int destination, int tag) {
  if (size_ == 0) return Error(XMPI_ERR_STATE, "mpi send: not initialised");
  if (destination < 0 || destination >= size_) return Error(XMPI_ERR_ARG, "mpi send: bad destination");
  Peer& P = *peers_[(size_t)destination];
  {
    std::lock_guard<std::mutex> g(P.mu);  // tagManager.Register (network.go:464-472), an error instead of a panic
    if (!P.send_tags.insert(tag).second) return Error(XMPI_ERR_TAG_EXISTS, "Tag " + std::to_string(tag) + " already in use sending");
  }
  // the user value -> bytes (network.go:537-542)
  gobwire::Bytes enc;
  switch (d.dtype) {
    case XMPI_U8: enc = gobwire::value_bytes((const uint8_t*)d.ptr, d.count, d.is_string); break;
    case XMPI_F64: enc = gobwire::value_slice((const double*)d.ptr, d.count, "[]float64", gobwire::kFloat); break;
    case XMPI_F32: enc = gobwire::value_slice((const float*)d.ptr, d.count, "[]float32", gobwire::kFloat); break;
    case XMPI_I64: enc = gobwire::value_slice((const int64_t*)d.ptr, d.count, "[]int64", gobwire::kInt); break;
    case XMPI_I32: enc = gobwire::value_slice((const int32_t*)d.ptr, d.count, "[]int32", gobwire::kInt); break;
    default: {
      std::lock_guard<std::mutex> g(P.mu);
      P.send_tags.erase(tag);
      return Error(XMPI_ERR_UNSUPPORTED, "mpi send: this payload type has no gob form here");
    }
  }
  Error result;
  if (destination == rank_) {  // local hand-off (network.go:545-548), still rendezvous: wait until it is taken
    std::unique_lock<std::mutex> l(P.mu);
    P.inbox[tag].push_back(std::move(enc));
    P.cv.notify_all();
    P.cv.wait(l, [&] { return P.acked.count(tag) > 0; });
    P.acked.erase(tag);
  } else {
    const gobwire::Bytes msg = gobwire::tagged_message(tag, enc.data(), enc.size());
    bool ok;
    {
      std::lock_guard<std::mutex> w(P.write_dial);  // one writer at a time per connection
      ok = write_all(P.dial_fd, msg.data(), msg.size());
    }
    std::unique_lock<std::mutex> l(P.mu);
    if (ok) P.cv.wait(l, [&] { return P.acked.count(tag) > 0 || P.acks_closed; });  // network.go:569
    if (!ok || !P.acked.count(tag)) result = Error(XMPI_ERR_PEER, "mpi send: connection to node " + std::to_string(destination) + " lost");
    P.acked.erase(tag);
  }
  std::lock_guard<std::mutex> g(P.mu);
  P.send_tags.erase(tag);  // network.go:571 (the reference forgets this for self-sends: quirk Q1)
  return result;
}

Error Network::Receive(Data d, int source, int tag) {
  if (size_ == 0) return Error(XMPI_ERR_STATE, "mpi receive: not initialised");
  if (source < 0 || source >= size_) return Error(XMPI_ERR_ARG, "mpi receive: bad source");
  Peer& P = *peers_[(size_t)source];
  gobwire::Bytes enc;
  {
    std::unique_lock<std::mutex> l(P.mu);
    if (!P.recv_tags.insert(tag).second) return Error(XMPI_ERR_TAG_EXISTS, "Tag " + std::to_string(tag) + " already in use receiving");
    P.cv.wait(l, [&] {
      auto it = P.inbox.find(tag);
      return (it != P.inbox.end() && !it->second.empty()) || P.data_closed;
    });
    auto it = P.inbox.find(tag);
    if (it == P.inbox.end() || it->second.empty()) {
      P.recv_tags.erase(tag);
      return Error(XMPI_ERR_PEER, "mpi receive: connection to node " + std::to_string(source) + " lost");
    }
    enc = std::move(it->second.front());
    it->second.pop_front();
    P.recv_tags.erase(tag);
    if (source == rank_) {  // release the local sender
      P.acked.insert(tag);
      P.cv.notify_all();
    }
  }
  if (source != rank_) {  // the ack: message{Tag} with no bytes, on the connection the data came in on (network.go:616-624)
    const gobwire::Bytes ack = gobwire::tagged_message(tag, nullptr, 0);
    std::lock_guard<std::mutex> w(P.write_listen);
    if (!write_all(P.listen_fd, ack.data(), ack.size())) return Error(XMPI_ERR_PEER, "mpi receive: cannot acknowledge");
  }
  // bytes -> the caller's storage (network.go:594-601); a slice is re-sized like gob's in-place decode
  gobwire::Reader body(nullptr, 0);
  gobwire::ValueHead h;
  if (!gobwire::open_value(enc.data(), enc.size(), &body, &h)) return Error(XMPI_ERR_ARG, "mpi receive: undecodable payload");
  const bool raw = h.type == gobwire::kByteSlice || h.type == gobwire::kString;
  if (raw != (d.dtype == XMPI_U8)) return Error(XMPI_ERR_ARG, "mpi receive: type of data differs from what was sent");
  const size_t n = (size_t)h.count;
  if (d.resize) d.resize(d.owner, n, &d);
  else if (d.count < n) return Error(XMPI_ERR_TRUNCATE, "mpi receive: message larger than the receive buffer");
  if (raw) {
    const uint8_t* q = body.take(n);
    if (!q) return Error(XMPI_ERR_ARG, "mpi receive: short payload");
    if (n) memcpy(d.ptr, q, n);
    return Error();
  }
  const bool floats = h.elem == gobwire::kFloat;
  if (floats != (d.dtype == XMPI_F64 || d.dtype == XMPI_F32)) return Error(XMPI_ERR_ARG, "mpi receive: type of data differs from what was sent");
  for (size_t k = 0; k < n; k++) {
    switch (d.dtype) {
      case XMPI_F64: ((double*)d.ptr)[k] = body.f(); break;
      case XMPI_F32: ((float*)d.ptr)[k] = (float)body.f(); break;  // exact: it was a float32 (gob widens)
      case XMPI_I64: ((int64_t*)d.ptr)[k] = body.i(); break;
      case XMPI_I32: ((int32_t*)d.ptr)[k] = (int32_t)body.i(); break;
      default: return Error(XMPI_ERR_UNSUPPORTED, "mpi receive: this payload type has no gob form here");
    }
  }
  return body.ok() ? Error() : Error(XMPI_ERR_ARG, "mpi receive: short payload");
}

// ---- collectives: whole-buffer exchange + rank-order fold on the host -----------------------------------------------
namespace {

constexpr int kCollTag = 0x7C011;  // the tag the collectives travel under (one collective at a time per job)

size_t elem_size(xmpi_dtype dt) {
  switch (dt) {
    case XMPI_U8: return 1;
    case XMPI_I32: case XMPI_F32: return 4;
    case XMPI_I64: case XMPI_F64: return 8;
    default: return 0;  // half / bfloat16 have no Go type: not on this backend
  }
}

template <typename T>
T fold1(T a, T b, xmpi_op op) {
  switch (op) {
    case XMPI_SUM: return (T)(a + b);
    case XMPI_PROD: return (T)(a * b);
    case XMPI_MIN: return (b < a) ? b : a;  // spelled like the kernels and the oracle: NaN / -0 behave alike
    default: return (a < b) ? b : a;
  }
}
template <typename T>
void fold_into(void* acc, const void* x, size_t n, xmpi_op op) {
  T* a = (T*)acc;
  const T* b = (const T*)x;
  for (size_t i = 0; i < n; i++) a[i] = fold1<T>(a[i], b[i], op);
}
// wrapping integer arithmetic (Go semantics) without signed-overflow UB
template <typename T, typename U>
void fold_into_int(void* acc, const void* x, size_t n, xmpi_op op) {
  T* a = (T*)acc;
  const T* b = (const T*)x;
  for (size_t i = 0; i < n; i++) {
    if (op == XMPI_SUM) a[i] = (T)((U)a[i] + (U)b[i]);
    else if (op == XMPI_PROD) a[i] = (T)((U)a[i] * (U)b[i]);
    else a[i] = fold1<T>(a[i], b[i], op);
  }
}

}  // namespace

// every rank sends its buffer to every rank (or to `only_to`) and receives everybody's: all[r] = rank r's bytes
Error Network::exchange(const Data& send, std::vector<std::vector<uint8_t>>* all, int only_to) {
  const size_t es = elem_size(send.dtype);
  if (!es) return Error(XMPI_ERR_UNSUPPORTED, "mpi collective: this element type has no Go counterpart on the TCP backend");
  const int n = size_;
  all->assign((size_t)n, std::vector<uint8_t>());
  std::vector<Error> serr((size_t)n), rerr((size_t)n);
  std::vector<std::thread> ts;
  for (int p = 0; p < n; p++)
    if (only_to < 0 || p == only_to) ts.emplace_back([&, p] { serr[(size_t)p] = Send(send, p, kCollTag); });
  if (only_to < 0 || only_to == rank_)
    for (int p = 0; p < n; p++)
      ts.emplace_back([&, p] {
        std::vector<uint8_t>& dst = (*all)[(size_t)p];
        Data d;
        d.dtype = send.dtype;
        d.is_string = send.is_string;
        d.owner = &dst;
        d.resize = [](void* o, size_t cnt, Data* self) {
          auto* v = static_cast<std::vector<uint8_t>*>(o);
          v->resize(cnt * elem_size(self->dtype));
          self->ptr = v->data();
          self->count = cnt;
        };
        rerr[(size_t)p] = Receive(d, p, kCollTag);
      });
  for (auto& t : ts) t.join();
  for (int p = 0; p < n; p++) {
    if (serr[(size_t)p]) return serr[(size_t)p];
    if (rerr[(size_t)p]) return rerr[(size_t)p];
  }
  return Error();
}

Error Network::Allgather(const Data& send, Data recv) {
  std::vector<std::vector<uint8_t>> all;
  if (Error e = exchange(send, &all, -1)) return e;
  const size_t es = elem_size(send.dtype), each = send.count * es;
  if (recv.resize) recv.resize(recv.owner, send.count * (size_t)size_, &recv);
  if (recv.count < send.count * (size_t)size_) return Error(XMPI_ERR_TRUNCATE, "mpi allgather: receive buffer too small");
  for (int r = 0; r < size_; r++) {
    if (all[(size_t)r].size() != each) return Error(XMPI_ERR_ARG, "mpi allgather: ranks passed different counts");
    if (each) memcpy((char*)recv.ptr + (size_t)r * each, all[(size_t)r].data(), each);
  }
  return Error();
}

Error Network::Reduce(const Data& send, Data recv, xmpi_op op, int root) {
  if (root < 0 || root >= size_) return Error(XMPI_ERR_ARG, "mpi reduce: bad root");
  std::vector<std::vector<uint8_t>> all;
  if (Error e = exchange(send, &all, root)) return e;
  if (rank_ != root) return Error();
  const size_t es = elem_size(send.dtype), bytes = send.count * es;
  if (recv.resize) recv.resize(recv.owner, send.count, &recv);
  if (recv.count < send.count) return Error(XMPI_ERR_TRUNCATE, "mpi reduce: receive buffer too small");
  for (int r = 0; r < size_; r++)
    if (all[(size_t)r].size() != bytes) return Error(XMPI_ERR_ARG, "mpi reduce: ranks passed different counts");
  if (bytes) memcpy(recv.ptr, all[0].data(), bytes);
  for (int r = 1; r < size_; r++) {  // strictly left to right in rank order: ((x0 op x1) op x2) ...
    const void* x = all[(size_t)r].data();
    switch (send.dtype) {
      case XMPI_U8: fold_into_int<uint8_t, uint8_t>(recv.ptr, x, send.count, op); break;
      case XMPI_I32: fold_into_int<int32_t, uint32_t>(recv.ptr, x, send.count, op); break;
      case XMPI_I64: fold_into_int<int64_t, uint64_t>(recv.ptr, x, send.count, op); break;
      case XMPI_F32: fold_into<float>(recv.ptr, x, send.count, op); break;
      case XMPI_F64: fold_into<double>(recv.ptr, x, send.count, op); break;
      default: return Error(XMPI_ERR_UNSUPPORTED, "mpi reduce: element type");
    }
  }
  return Error();
}

Error Network::Allreduce(const Data& send, Data recv, xmpi_op op) {
  // every rank gathers everything and folds for itself -- what the oracle's definition says, not the cheapest schedule
  std::vector<std::vector<uint8_t>> all;
  if (Error e = exchange(send, &all, -1)) return e;
  const size_t es = elem_size(send.dtype), bytes = send.count * es;
  if (recv.resize) recv.resize(recv.owner, send.count, &recv);
  if (recv.count < send.count) return Error(XMPI_ERR_TRUNCATE, "mpi allreduce: receive buffer too small");
  for (int r = 0; r < size_; r++)
    if (all[(size_t)r].size() != bytes) return Error(XMPI_ERR_ARG, "mpi allreduce: ranks passed different counts");
  if (bytes) memcpy(recv.ptr, all[0].data(), bytes);
  for (int r = 1; r < size_; r++) {
    const void* x = all[(size_t)r].data();
    switch (send.dtype) {
      case XMPI_U8: fold_into_int<uint8_t, uint8_t>(recv.ptr, x, send.count, op); break;
      case XMPI_I32: fold_into_int<int32_t, uint32_t>(recv.ptr, x, send.count, op); break;
      case XMPI_I64: fold_into_int<int64_t, uint64_t>(recv.ptr, x, send.count, op); break;
      case XMPI_F32: fold_into<float>(recv.ptr, x, send.count, op); break;
      case XMPI_F64: fold_into<double>(recv.ptr, x, send.count, op); break;
      default: return Error(XMPI_ERR_UNSUPPORTED, "mpi allreduce: element type");
    }
  }
  return Error();
}

Error Network::Bcast(Data buf, int root) {
  if (root < 0 || root >= size_) return Error(XMPI_ERR_ARG, "mpi bcast: bad root");
  if (rank_ == root) {
    std::vector<Error> errs((size_t)size_);
    std::vector<std::thread> ts;
    for (int p = 0; p < size_; p++)
      if (p != root) ts.emplace_back([&, p] { errs[(size_t)p] = Send(buf, p, kCollTag); });
    for (auto& t : ts) t.join();
    for (const Error& e : errs)
      if (e) return e;
    return Error();
  }
  return Receive(buf, root, kCollTag);
}

Error Network::Barrier() {
  std::vector<uint8_t> token = {1}, all;
  return Allgather(Slice(token), Into(&all));
}

void Network::close_all() {
  for (auto& up : peers_) {
    Peer& P = *up;
    if (P.dial_fd >= 0) ::shutdown(P.dial_fd, SHUT_RDWR);
    if (P.listen_fd >= 0) ::shutdown(P.listen_fd, SHUT_RDWR);
    if (P.data_reader.joinable()) P.data_reader.join();
    if (P.ack_reader.joinable()) P.ack_reader.join();
    if (P.dial_fd >= 0) ::close(P.dial_fd);
    if (P.listen_fd >= 0) ::close(P.listen_fd);
    P.dial_fd = P.listen_fd = -1;
  }
}

void Network::Finalize() {  // network.go:354-369
  close_all();
  peers_.clear();
  size_ = 0;
}

}  // namespace mpi
