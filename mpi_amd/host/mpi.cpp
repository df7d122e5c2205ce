// mpi.cpp -- C++ mirror of the reference's Go package `mpi` on top of the C ABI (see mpi.hpp).
#include "mpi.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace mpi {

// ---- flags.go ------------------------------------------------------------------------------------
std::string FlagAddr;
std::vector<std::string> FlagAllAddrs;
double FlagInitTimeout = 0;
std::string FlagProtocol = "tcp";  // flags.go:48 default; this backend ignores it
std::string FlagPassword;

static double parse_duration(const std::string& s) {  // time.ParseDuration subset: 1.5s, 200ms, 2m, 1h
  double total = 0;
  size_t i = 0;
  while (i < s.size()) {
    size_t j = i;
    while (j < s.size() && (isdigit((unsigned char)s[j]) || s[j] == '.')) j++;
    double v = atof(s.substr(i, j - i).c_str());
    size_t k = j;
    while (k < s.size() && isalpha((unsigned char)s[k])) k++;
    const std::string u = s.substr(j, k - j);
    if (u == "ns") v *= 1e-9;
    else if (u == "us") v *= 1e-6;
    else if (u == "ms") v *= 1e-3;
    else if (u == "s" || u.empty()) v *= 1;
    else if (u == "m") v *= 60;
    else if (u == "h") v *= 3600;
    else throw std::invalid_argument("time: unknown unit in duration " + s);
    total += v;
    i = k;
  }
  return total;
}

void ParseFlags(int* argc, char** argv) {
  int out = 1;
  for (int i = 1; i < *argc; i++) {
    std::string a = argv[i];
    std::string name, val;
    bool has_val = false;
    if (a.size() > 1 && a[0] == '-') {
      size_t start = (a[1] == '-') ? 2 : 1;
      size_t eq = a.find('=');
      name = a.substr(start, eq == std::string::npos ? std::string::npos : eq - start);
      if (eq != std::string::npos) {
        val = a.substr(eq + 1);
        has_val = true;
      }
    }
    const bool ours = name == "mpi-addr" || name == "mpi-alladdr" || name == "mpi-inittimeout" ||
                      name == "mpi-protocol" || name == "mpi-password";
    if (!ours) {
      argv[out++] = argv[i];
      continue;
    }
    if (!has_val) {
      if (i + 1 >= *argc) throw std::invalid_argument("flag needs an argument: -" + name);
      val = argv[++i];
    }
    if (name == "mpi-addr") FlagAddr = val;
    else if (name == "mpi-alladdr") {  // AddrsFlag.Set appends (flags.go:22-27)
      size_t p = 0;
      while (p <= val.size()) {
        size_t q = val.find(',', p);
        if (q == std::string::npos) q = val.size();
        FlagAllAddrs.push_back(val.substr(p, q - p));
        p = q + 1;
      }
    } else if (name == "mpi-inittimeout") FlagInitTimeout = parse_duration(val);
    else if (name == "mpi-protocol") FlagProtocol = val;
    else FlagPassword = val;
  }
  *argc = out;
}

// ---- mpi.go: registry + package-level delegates ------------------------------------------------------
static XGMI g_default;
static Interface* mpier = &g_default;
static bool registerCalled = false;

XGMI* DefaultBackend() { return &g_default; }

void Register(Interface* impl) {
  mpier = impl;
  if (registerCalled) throw std::logic_error("register called more than once");
  registerCalled = true;
}

Error Init() { return mpier->Init(); }
void Finalize() { mpier->Finalize(); }
int Rank() { return mpier->Rank(); }
int Size() { return mpier->Size(); }
Error Send(const Data& data, int destination, int tag) { return mpier->Send(data, destination, tag); }
Error Receive(Data data, int source, int tag) { return mpier->Receive(data, source, tag); }

static Collective* collective_or_null() { return dynamic_cast<Collective*>(mpier); }
static Error no_collectives() { return Error(XMPI_ERR_UNSUPPORTED, "the registered mpi backend implements no collectives"); }

Error Bcast(Data buf, int root) {
  Collective* c = collective_or_null();
  return c ? c->Bcast(buf, root) : no_collectives();
}
Error Reduce(const Data& send, Data recv, xmpi_op op, int root) {
  Collective* c = collective_or_null();
  return c ? c->Reduce(send, recv, op, root) : no_collectives();
}
Error Allreduce(const Data& send, Data recv, xmpi_op op) {
  Collective* c = collective_or_null();
  return c ? c->Allreduce(send, recv, op) : no_collectives();
}
Error Allgather(const Data& send, Data recv) {
  Collective* c = collective_or_null();
  return c ? c->Allgather(send, recv) : no_collectives();
}
Error Barrier() {
  Collective* c = collective_or_null();
  return c ? c->Barrier() : no_collectives();
}

// ---- the xGMI backend ----------------------------------------------------------------------------------
static Error from_code(int rc, const char* where) {
  if (rc == XMPI_OK) return Error();
  if (rc == XMPI_ERR_TAG_EXISTS) return Error(rc, std::string(xmpi_last_error()));  // TagExists.Error(), mpi.go:180-182
  return Error(rc, std::string(where) + ": " + xmpi_strerror(rc) + (*xmpi_last_error() ? std::string("; ") + xmpi_last_error() : ""));
}

static uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull) {
  for (unsigned char ch : s) {
    h ^= ch;
    h *= 1099511628211ull;
  }
  return h;
}

Error XGMI::Init() {
  // useFlags (network.go:69-90): zero-valued fields come from the flags
  if (Password.empty()) Password = FlagPassword;
  if (Timeout == 0) Timeout = FlagInitTimeout;
  if (Addr.empty()) Addr = FlagAddr;
  if (Addrs.empty()) Addrs = FlagAllAddrs;
  if (Addrs.empty()) {  // network.go:55-58: a single node
    Addr = ":5000";
    Addrs = {":5000"};
  }
  // assignRanks (network.go:94-109): lexicographic sort, uniqueness, rank = index of the own address
  std::sort(Addrs.begin(), Addrs.end());
  for (size_t i = 0; i + 1 < Addrs.size(); i++)
    if (Addrs[i] == Addrs[i + 1]) return Error(XMPI_ERR_ARG, "network addresses not unique");
  auto it = std::lower_bound(Addrs.begin(), Addrs.end(), Addr);
  if (it == Addrs.end() || *it != Addr)
    return Error(XMPI_ERR_ARG, "mpi init: local ip address not in global list. Local address is: " + Addr);
  const int rank = (int)(it - Addrs.begin());
  const int size = (int)Addrs.size();
  // every rank derives the same rendezvous key: the launcher's job id when there is one, and the
  // address list + password (ranks with a different password never meet: network.go:343-346)
  std::string all;
  for (const std::string& a : Addrs) all += a + ",";
  const char* job = getenv("XMPI_JOB");
  char key[96];
  snprintf(key, sizeof key, "%s%016llx", job ? job : "j", (unsigned long long)fnv1a(Password, fnv1a(all)));
  int device = Device;
  if (device < 0 && getenv("XMPI_DEVICE")) device = atoi(getenv("XMPI_DEVICE"));
  if (Timeout > 0) {
    char t[32];
    snprintf(t, sizeof t, "%d", (int)(Timeout + 0.999));
    setenv("XMPI_INIT_TIMEOUT_S", t, 1);  // bounds Init only, as -mpi-inittimeout does (network.go:223-234,307-312)
  }
  return from_code(xmpi_init(rank, size, device, key, &comm_), "mpi init");
}

void XGMI::Finalize() {
  if (comm_) xmpi_finalize(comm_);
  comm_ = nullptr;
}

int XGMI::Rank() { return xmpi_rank(comm_); }  // -1 before Init (network.go:41-46)
int XGMI::Size() { return xmpi_size(comm_); }  //  0 before Init (network.go:48-50)

Error XGMI::Send(const Data& d, int destination, int tag) {
  return from_code(xmpi_send(comm_, d.ptr, d.count, d.dtype, destination, tag), "mpi send");
}

Error XGMI::SendNoWait(const Data& d, int destination, int tag) {
  return from_code(xmpi_send_nowait(comm_, d.ptr, d.count, d.dtype, destination, tag), "mpi send");
}

Error XGMI::Wait(int destination, int tag) { return from_code(xmpi_wait(comm_, destination, tag), "mpi wait"); }

Error XGMI::Receive(Data d, int source, int tag) {
  if (d.resize) {  // decode into *[]T: size the container to the incoming message first
    size_t n = 0;
    xmpi_dtype dt = XMPI_U8;
    int rc = xmpi_probe(comm_, source, tag, &n, &dt);
    if (rc != XMPI_OK) return from_code(rc, "mpi receive");
    if (dt != d.dtype) return Error(XMPI_ERR_ARG, "mpi receive: type of data differs from what was sent");
    d.resize(d.owner, n, &d);
  }
  size_t got = 0;
  return from_code(xmpi_recv(comm_, d.ptr, d.count, d.dtype, source, tag, &got), "mpi receive");
}

Error XGMI::Bcast(Data buf, int root) { return from_code(xmpi_bcast(comm_, buf.ptr, buf.count, buf.dtype, root, XMPI_ALGO_AUTO), "mpi bcast"); }

Error XGMI::Reduce(const Data& send, Data recv, xmpi_op op, int root) {
  return from_code(xmpi_reduce(comm_, send.ptr, recv.ptr, send.count, send.dtype, op, root, XMPI_ALGO_AUTO), "mpi reduce");
}

Error XGMI::Allreduce(const Data& send, Data recv, xmpi_op op) {
  if (recv.resize) recv.resize(recv.owner, send.count, &recv);
  return from_code(xmpi_allreduce(comm_, send.ptr, recv.ptr, send.count, send.dtype, op, Algo), "mpi allreduce");
}

Error XGMI::Allgather(const Data& send, Data recv) {
  if (recv.resize) recv.resize(recv.owner, send.count * (size_t)Size(), &recv);
  return from_code(xmpi_allgather(comm_, send.ptr, recv.ptr, send.count, send.dtype, Algo == XMPI_ALGO_RHD ? XMPI_ALGO_AUTO : Algo),
                   "mpi allgather");
}

Error XGMI::Barrier() { return from_code(xmpi_barrier(comm_), "mpi barrier"); }

Error XGMI::IAllreduce(const Data& send, Data recv, xmpi_op op, xmpi_request** req) {
  if (recv.resize) recv.resize(recv.owner, send.count, &recv);
  return from_code(xmpi_iallreduce(comm_, send.ptr, recv.ptr, send.count, send.dtype, op, Algo, req), "mpi iallreduce");
}

void* XGMI::Stream() { return xmpi_stream_create(comm_); }
void XGMI::StreamDestroy(void* stream) { xmpi_stream_destroy(comm_, stream); }
Error XGMI::StreamSync(void* stream) { return from_code(xmpi_stream_sync(comm_, stream), "mpi stream sync"); }
Error XGMI::AllreduceOnStream(const Data& send, Data recv, xmpi_op op, void* stream) {
  return from_code(xmpi_allreduce_on_stream(comm_, send.ptr, recv.ptr, send.count, send.dtype, op, stream), "mpi allreduce");
}
Error XGMI::AllgatherOnStream(const Data& send, Data recv, void* stream) {
  return from_code(xmpi_allgather_on_stream(comm_, send.ptr, recv.ptr, send.count, send.dtype, stream), "mpi allgather");
}
Error XGMI::BcastOnStream(Data buf, int root, void* stream) {
  return from_code(xmpi_bcast_on_stream(comm_, buf.ptr, buf.count, buf.dtype, root, stream), "mpi bcast");
}
Error XGMI::ReduceOnStream(const Data& send, Data recv, xmpi_op op, int root, void* stream) {
  return from_code(xmpi_reduce_on_stream(comm_, send.ptr, recv.ptr, send.count, send.dtype, op, root, stream), "mpi reduce");
}

Error XGMI::GraphBegin(void* stream) { return from_code(xmpi_graph_begin(comm_, stream), "mpi graph begin"); }
Error XGMI::GraphEnd(void* stream, void** graph) { return from_code(xmpi_graph_end(comm_, stream, graph), "mpi graph end"); }
Error XGMI::GraphLaunch(void* graph, void* stream) { return from_code(xmpi_graph_launch(comm_, graph, stream), "mpi graph launch"); }
void XGMI::GraphDestroy(void* graph) { xmpi_graph_destroy(comm_, graph); }

Error XGMI::WaitRequest(xmpi_request* req) { return from_code(xmpi_request_wait(req), "mpi wait"); }

Error XGMI::RegisterBuffer(void* p, size_t bytes) { return from_code(xmpi_register(comm_, p, bytes), "mpi register"); }
Error XGMI::DeregisterBuffer(void* p) { return from_code(xmpi_deregister(comm_, p), "mpi deregister"); }

void* XGMI::Malloc(size_t bytes) { return xmpi_malloc(comm_, bytes); }
void XGMI::Free(void* p) { xmpi_free(comm_, p); }
Error XGMI::Memcpy(void* dst, const void* src, size_t bytes) { return from_code(xmpi_memcpy(comm_, dst, src, bytes), "mpi memcpy"); }

}  // namespace mpi
