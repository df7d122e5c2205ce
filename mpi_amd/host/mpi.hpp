// mpi.hpp -- C++ mirror of the reference's Go package `mpi` (host side of the drop-in boundary).
//
// The reference is compiled Go; this image has no Go toolchain, so the host side above the C ABI
// (include/xmpi.h) is written in C++ with the reference's names, argument meaning and error
// behaviour.  The Go binding a maintainer would add is in go/xgmi/ and INTEGRATION.md.
//
//   reference (Go)                                   here (C++)
//   mpi.Register(mpi.Interface)      mpi.go:61-67    mpi::Register(Interface*)   (2nd call throws: Go panics)
//   mpi.Init() error                 mpi.go:96-98    mpi::Init() -> Error
//   mpi.Finalize()                   mpi.go:102-104  mpi::Finalize()
//   mpi.Rank() / mpi.Size()          mpi.go:112-119  mpi::Rank() / mpi::Size()   (-1 / 0 before Init)
//   mpi.Send(data, dest, tag)        mpi.go:126-128  mpi::Send(data, dest, tag)
//   mpi.Receive(&data, src, tag)     mpi.go:157-159  mpi::Receive(&data, src, tag)
//   type Interface                   mpi.go:163-170  class Interface
//   type Raw []byte                  mpi.go:75-91    using Raw = std::vector<uint8_t>
//   type TagExists                   mpi.go:172-182  Error::IsTagExists()
//   flags -mpi-addr ...              flags.go:44-50  mpi::ParseFlags(&argc, argv)
//   type Network (TCP backend)       network.go      class XGMI (HBM windows over xGMI, via libxmpi.so);
//                                                    class Network (network.hpp): the reference's own TCP + gob
//                                                    protocol, wire-compatible, for CPU / mixed / multi-node ranks
//   //func AllReduce() {}            mpi.go:130      mpi::Allreduce / Bcast / Reduce / Allgather
//                                                    (optional Collective interface, cf. the unused
//                                                     isAllReducer probe at mpi.go:69-71)
// `data interface{}` becomes a typed view (Data) over host or HBM memory: the reference types the
// payload by reflection through gob (network.go:539,597); a device buffer needs an explicit tag.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "xmpi.h"

namespace mpi {

// Go's `error`: empty message == nil
class Error {
 public:
  Error() = default;
  Error(int code, std::string msg) : code_(code), msg_(std::move(msg)) {}
  explicit operator bool() const { return code_ != 0; }  // true == an error occurred
  int Code() const { return code_; }
  const std::string& What() const { return msg_; }
  bool IsTagExists() const { return code_ == XMPI_ERR_TAG_EXISTS; }  // mpi.go:172-182

 private:
  int code_ = 0;
  std::string msg_;
};

using Raw = std::vector<uint8_t>;  // mpi.go:75: sent as raw bytes, no encoding

// What `data interface{}` carries on this backend: a typed span of host memory or of this rank's HBM.
struct Data {
  void* ptr = nullptr;
  size_t count = 0;
  xmpi_dtype dtype = XMPI_U8;
  bool is_string = false;  // a Go `string` rather than a `[]byte` (they differ on the reference's wire: gob ids 6 / 5)
  // Receive only: the container to re-size to the incoming length (Go re-slices / re-allocates the
  // pointed-to slice, mpi.go:83-91, bounce.go:89,94).  Null for fixed spans (device buffers).
  void* owner = nullptr;
  void (*resize)(void* owner, size_t count, Data* self) = nullptr;
};

template <typename T> struct DTypeOf;
template <> struct DTypeOf<uint8_t> { static constexpr xmpi_dtype v = XMPI_U8; };
template <> struct DTypeOf<char> { static constexpr xmpi_dtype v = XMPI_U8; };
template <> struct DTypeOf<int32_t> { static constexpr xmpi_dtype v = XMPI_I32; };
template <> struct DTypeOf<int64_t> { static constexpr xmpi_dtype v = XMPI_I64; };
template <> struct DTypeOf<float> { static constexpr xmpi_dtype v = XMPI_F32; };
template <> struct DTypeOf<double> { static constexpr xmpi_dtype v = XMPI_F64; };
struct Float16 { uint16_t bits; };  // Go has no float16 either: `type Float16 uint16`
template <> struct DTypeOf<Float16> { static constexpr xmpi_dtype v = XMPI_F16; };

// Send side views
template <typename T> Data Slice(const std::vector<T>& v) {
  Data d; d.ptr = const_cast<T*>(v.data()); d.count = v.size(); d.dtype = DTypeOf<T>::v; return d;
}
inline Data Slice(const std::string& s) {
  Data d; d.ptr = const_cast<char*>(s.data()); d.count = s.size(); d.dtype = XMPI_U8; d.is_string = true; return d;
}
template <typename T> Data Span(const T* p, size_t n) {  // host or device pointer
  Data d; d.ptr = const_cast<T*>(p); d.count = n; d.dtype = DTypeOf<T>::v; return d;
}
// Receive side views: the container grows / shrinks to the message (like decoding into *[]T)
template <typename T> Data Into(std::vector<T>* v) {
  Data d; d.ptr = v->data(); d.count = v->size(); d.dtype = DTypeOf<T>::v; d.owner = v;
  d.resize = [](void* o, size_t n, Data* self) { auto* vv = static_cast<std::vector<T>*>(o); vv->resize(n); self->ptr = vv->data(); self->count = n; };
  return d;
}
inline Data Into(std::string* s) {
  Data d; d.ptr = &(*s)[0]; d.count = s->size(); d.dtype = XMPI_U8; d.is_string = true; d.owner = s;
  d.resize = [](void* o, size_t n, Data* self) { auto* ss = static_cast<std::string*>(o); ss->resize(n); self->ptr = &(*ss)[0]; self->count = n; };
  return d;
}

// mpi.go:163-170
class Interface {
 public:
  virtual ~Interface() = default;
  virtual Error Init() = 0;
  virtual void Finalize() = 0;
  virtual int Rank() = 0;
  virtual int Size() = 0;
  virtual Error Send(const Data& data, int destination, int tag) = 0;
  virtual Error Receive(Data data, int source, int tag) = 0;
};

// The collectives the reference only stubs (mpi.go:130).  A backend may implement them; the
// package-level functions probe for this interface (cf. `isAllReducer`, mpi.go:69-71).
class Collective {
 public:
  virtual ~Collective() = default;
  virtual Error Bcast(Data buf, int root) = 0;
  virtual Error Reduce(const Data& send, Data recv, xmpi_op op, int root) = 0;
  virtual Error Allreduce(const Data& send, Data recv, xmpi_op op) = 0;
  virtual Error Allgather(const Data& send, Data recv) = 0;
  virtual Error Barrier() = 0;
};

// flags.go:10-14
extern std::string FlagAddr;
extern std::vector<std::string> FlagAllAddrs;
extern double FlagInitTimeout;  // seconds (Go: time.Duration)
extern std::string FlagProtocol;
extern std::string FlagPassword;
// flag.Parse() for the five -mpi-* flags ("-mpi-addr :6000", "-mpi-addr=:6000", "--mpi-addr ...");
// recognised flags are removed from argv, everything else is left for the program.
void ParseFlags(int* argc, char** argv);

// mpi.go:61-67: call once, before Init; a second call throws std::logic_error ("register called
// more than once" -- the reference panics).
void Register(Interface* impl);

Error Init();
void Finalize();
int Rank();
int Size();
Error Send(const Data& data, int destination, int tag);
Error Receive(Data data, int source, int tag);

Error Bcast(Data buf, int root);
Error Reduce(const Data& send, Data recv, xmpi_op op, int root);
Error Allreduce(const Data& send, Data recv, xmpi_op op = XMPI_SUM);
Error Allgather(const Data& send, Data recv);
Error Barrier();

// The MI355X backend: replaces type Network (network.go:25-39).  Zero-valued fields are taken from
// the flags (network.go:69-90); struct fields win over flags.
class XGMI : public Interface, public Collective {
 public:
  std::string Addr;                // -mpi-addr: this process's entry in Addrs (decides the rank)
  std::vector<std::string> Addrs;  // -mpi-alladdr: rank = index in the lexicographically sorted list
  double Timeout = 0;              // -mpi-inittimeout, seconds
  std::string Password;            // -mpi-password: ranks with different passwords do not meet
  int Device = -1;                 // GPU ordinal; -1 = $XMPI_DEVICE, else rank % visible GPUs
  int Algo = XMPI_ALGO_AUTO;       // schedule for the collectives

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

  // The Send / Wait pair the reference sketches in a comment (mpi.go:132-152): SendNoWait returns once
  // the payload has left `data`; Wait blocks until `destination` confirmed the message with `tag` and
  // frees the {destination, tag} pair.
  Error SendNoWait(const Data& data, int destination, int tag);
  Error Wait(int destination, int tag);

  // Non-blocking allreduce: returns at once, WaitRequest blocks until it completed (operations issued
  // this way run in issue order on the communicator's worker; the buffers belong to the operation until then).
  Error IAllreduce(const Data& send, Data recv, xmpi_op op, xmpi_request** req);
  Error WaitRequest(xmpi_request* req);

  // Stream-ordered collectives (xmpi_*_on_stream): enqueued on a HIP stream like a kernel launch, no host
  // wait; Stream() = a new stream of this rank's GPU (nullptr = the communicator's own), StreamSync waits for
  // everything enqueued on it.  The overlap the reference sketched for Send / Wait (mpi.go:132-152), on streams.
  void* Stream();
  void StreamDestroy(void* stream);
  Error StreamSync(void* stream);
  Error AllreduceOnStream(const Data& send, Data recv, xmpi_op op, void* stream);
  Error AllgatherOnStream(const Data& send, Data recv, void* stream);
  Error BcastOnStream(Data buf, int root, void* stream);
  Error ReduceOnStream(const Data& send, Data recv, xmpi_op op, int root, void* stream);
  // hipGraph capture of what is enqueued on `stream` between GraphBegin and GraphEnd; GraphLaunch replays it
  Error GraphBegin(void* stream);
  Error GraphEnd(void* stream, void** graph);
  Error GraphLaunch(void* graph, void* stream);
  void GraphDestroy(void* graph);

  // Device memory from another allocator joins the zero-copy paths with RegisterBuffer (Malloc'd memory
  // is registered as it is); DeregisterBuffer before it is freed.
  Error RegisterBuffer(void* p, size_t bytes);
  Error DeregisterBuffer(void* p);

  // HBM buffers of this rank (device-resident payloads are the hot path)
  void* Malloc(size_t bytes);
  void Free(void* p);
  Error Memcpy(void* dst, const void* src, size_t bytes);
  xmpi_comm* Handle() { return comm_; }

 private:
  xmpi_comm* comm_ = nullptr;
};

// the default backend (the reference's default is &Network{}, mpi.go:56)
XGMI* DefaultBackend();

}  // namespace mpi
