// gobwire.hpp -- the slice of Go's encoding/gob wire format that btracey/mpi's TCP backend puts on its
// connections, for mpi::Network (network.hpp): a product-side implementation, independent of the test oracle
// (oracle/gob_codec.h is test infrastructure and is not included, linked or executed from here).
//
// What the reference sends (network.go; every Encode uses a FRESH gob.Encoder, so type descriptors precede
// every value):
//   handshake   initialMessage{Password string; Id int}            network.go:198-201,242,258,318,328
//   data / ack  message{Tag int; Bytes Raw}                         network.go:511-514,562,609,620
//               Raw implements GobEncoder (mpi.go:75-91): its field travels as a length-prefixed byte string
//   payload     Bytes = the user value encoded by another fresh encoder (network.go:539): []byte, string,
//               []float64, []float32 (widened to float64), []int64, []int32
// Format rules (package encoding/gob, "Encoding Details"): an unsigned integer below 128 is one byte, otherwise
// a byte holding the negated byte count followed by the big-endian bytes; a signed integer is folded into an
// unsigned one with the sign in bit 0; a float is its IEEE-754 float64 image byte-reversed, sent as unsigned;
// a stream is a sequence of messages, each = unsigned length + body; a body starts with a signed type id --
// negative: this message defines that type (a wireType struct follows), positive: a value of that type;
// struct fields are sent as (field number delta, value) pairs, zero values omitted, terminated by a 0;
// a top-level non-struct value is preceded by one 0 byte.  User type ids start at 65, in order of first use.
// No Go toolchain exists in this image: interoperability is tested against the repository's restatement of the
// reference (tests/test_tcp_backend.py drives oracle/refpath_bin as the peer), and the primitives against the
// known answers of gob's format document.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace mpi {
namespace gobwire {

using Bytes = std::vector<uint8_t>;

enum BuiltinId : int { kBool = 1, kInt = 2, kUint = 3, kFloat = 4, kByteSlice = 5, kString = 6 };
constexpr int kFirstUserType = 65;

// ---- writer ------------------------------------------------------------------------------------------
class Writer {
 public:
  Bytes out;

  void u(uint64_t v) {
    if (v < 0x80) {
      out.push_back((uint8_t)v);
      return;
    }
    int n = 8;
    while (n > 1 && ((v >> (8 * (n - 1))) & 0xFF) == 0) n--;
    out.push_back((uint8_t)(0x100 - n));
    for (int k = n - 1; k >= 0; k--) out.push_back((uint8_t)(v >> (8 * k)));
  }
  void i(int64_t v) { u(v < 0 ? ((uint64_t)(~v) << 1) | 1u : (uint64_t)v << 1); }
  void f(double d) {
    uint64_t bits;
    memcpy(&bits, &d, 8);
    u(__builtin_bswap64(bits));
  }
  void str(const char* p, size_t n) {
    u(n);
    out.insert(out.end(), (const uint8_t*)p, (const uint8_t*)p + n);
  }
  void str(const std::string& s) { str(s.data(), s.size()); }
};

// one message: length prefix + body
inline void put_message(Bytes* stream, const Writer& body) {
  Writer len;
  len.u(body.out.size());
  stream->insert(stream->end(), len.out.begin(), len.out.end());
  stream->insert(stream->end(), body.out.begin(), body.out.end());
}

// wireType is a struct {ArrayT, SliceT, StructT, MapT, GobEncoderT ...}: exactly one field is set; each of those
// starts with CommonType{Name, Id}
inline void common(Writer& w, const std::string& name, int id) {
  w.u(1);  // CommonType.Name
  w.str(name);
  w.u(1);  // CommonType.Id
  w.i(id);
  w.u(0);
}

inline void define_slice(Bytes* stream, int id, const std::string& name, int elem) {
  Writer w;
  w.i(-id);
  w.u(2);  // wireType.SliceT (field 1, delta from -1)
  w.u(1);  // sliceType.CommonType
  common(w, name, id);
  w.u(1);  // sliceType.Elem
  w.i(elem);
  w.u(0);
  w.u(0);
  put_message(stream, w);
}

struct Field {
  const char* name;
  int type;
};

inline void define_struct(Bytes* stream, int id, const std::string& name, const Field* fields, size_t nfields) {
  Writer w;
  w.i(-id);
  w.u(3);  // wireType.StructT (field 2)
  w.u(1);  // structType.CommonType
  common(w, name, id);
  w.u(1);  // structType.Field
  w.u(nfields);
  for (size_t k = 0; k < nfields; k++) {
    w.u(1);
    w.str(fields[k].name, strlen(fields[k].name));
    w.u(1);
    w.i(fields[k].type);
    w.u(0);
  }
  w.u(0);
  w.u(0);
  put_message(stream, w);
}

inline void define_gobencoder(Bytes* stream, int id, const std::string& name) {
  Writer w;
  w.i(-id);
  w.u(5);  // wireType.GobEncoderT (field 4)
  w.u(1);
  common(w, name, id);
  w.u(0);
  w.u(0);
  put_message(stream, w);
}

// ---- what the reference sends --------------------------------------------------------------------------
inline Bytes initial_message(const std::string& password, int64_t id) {
  static const Field fields[] = {{"Password", kString}, {"Id", kInt}};
  Bytes s;
  define_struct(&s, kFirstUserType, "initialMessage", fields, 2);
  Writer w;
  w.i(kFirstUserType);
  int last = -1;
  if (!password.empty()) {
    w.u((uint64_t)(0 - last));
    w.str(password);
    last = 0;
  }
  if (id != 0) {
    w.u((uint64_t)(1 - last));
    w.i(id);
  }
  w.u(0);
  put_message(&s, w);
  return s;
}

inline Bytes tagged_message(int64_t tag, const uint8_t* payload, size_t n) {
  static const Field fields[] = {{"Tag", kInt}, {"Bytes", kFirstUserType + 1}};
  Bytes s;
  s.reserve(n + 128);
  define_struct(&s, kFirstUserType, "message", fields, 2);
  define_gobencoder(&s, kFirstUserType + 1, "Raw");
  Writer w;
  w.out.reserve(n + 24);
  w.i(kFirstUserType);
  int last = -1;
  if (tag != 0) {
    w.u((uint64_t)(0 - last));
    w.i(tag);
    last = 0;
  }
  if (n != 0) {
    w.u((uint64_t)(1 - last));
    w.str((const char*)payload, n);
  }
  w.u(0);
  put_message(&s, w);
  return s;
}

// top-level user values (network.go:539)
inline Bytes value_bytes(const uint8_t* p, size_t n, bool as_string) {
  Bytes s;
  Writer w;
  w.out.reserve(n + 16);
  w.i(as_string ? kString : kByteSlice);
  w.u(0);
  w.str((const char*)p, n);
  put_message(&s, w);
  return s;
}
template <typename T>
inline Bytes value_slice(const T* v, size_t n, const char* go_name, int elem) {
  Bytes s;
  define_slice(&s, kFirstUserType, go_name, elem);
  Writer w;
  w.out.reserve(n * 9 + 16);
  w.i(kFirstUserType);
  w.u(0);
  w.u(n);
  for (size_t k = 0; k < n; k++) {
    if (elem == kFloat) w.f((double)v[k]);
    else w.i((int64_t)v[k]);
  }
  put_message(&s, w);
  return s;
}

// ---- reader --------------------------------------------------------------------------------------------
class Reader {
 public:
  Reader(const uint8_t* p, size_t n) : p_(p), e_(p + n) {}
  bool ok() const { return ok_; }
  bool at_end() const { return p_ >= e_; }
  size_t consumed(const uint8_t* start) const { return (size_t)(p_ - start); }
  uint64_t u() {
    if (p_ >= e_) return fail();
    const uint8_t c = *p_++;
    if (c < 0x80) return c;
    const int n = 0x100 - (int)c;
    if (n > 8 || e_ - p_ < n) return fail();
    uint64_t v = 0;
    for (int k = 0; k < n; k++) v = (v << 8) | *p_++;
    return v;
  }
  int64_t i() {
    const uint64_t v = u();
    return (v & 1) ? (int64_t)~(v >> 1) : (int64_t)(v >> 1);
  }
  double f() {
    const uint64_t bits = __builtin_bswap64(u());
    double d;
    memcpy(&d, &bits, 8);
    return d;
  }
  const uint8_t* take(size_t n) {
    if ((size_t)(e_ - p_) < n) {
      fail();
      return nullptr;
    }
    const uint8_t* q = p_;
    p_ += n;
    return q;
  }
  // the next message of the stream: its body as a sub-reader
  bool message(Reader* body) {
    const uint64_t len = u();
    const uint8_t* q = ok_ ? take((size_t)len) : nullptr;
    if (!q) return false;
    *body = Reader(q, (size_t)len);
    return true;
  }
  // skip type definitions; *id = the type id of the value message, body positioned behind it
  bool value(Reader* body, int64_t* id) {
    while (message(body)) {
      *id = body->i();
      if (!body->ok()) return false;
      if (*id > 0) return true;
    }
    return false;
  }

 private:
  uint64_t fail() {
    ok_ = false;
    return 0;
  }
  const uint8_t* p_;
  const uint8_t* e_;
  bool ok_ = true;
};

// struct {first: string/int ...}: calls on_field(field number, reader) for every field present
template <typename F>
inline bool struct_fields(Reader& body, F on_field) {
  int field = -1;
  for (;;) {
    const uint64_t delta = body.u();
    if (!body.ok()) return false;
    if (delta == 0) return true;
    field += (int)delta;
    if (!on_field(field, body)) return false;
  }
}

inline bool parse_initial(const uint8_t* p, size_t n, std::string* password, int64_t* id) {
  Reader r(p, n), body(nullptr, 0);
  int64_t type = 0;
  if (!r.value(&body, &type)) return false;
  password->clear();
  *id = 0;
  return struct_fields(body, [&](int field, Reader& b) {
    if (field == 0) {
      const uint64_t len = b.u();
      const uint8_t* q = b.ok() ? b.take((size_t)len) : nullptr;
      if (!q) return false;
      password->assign((const char*)q, (size_t)len);
      return true;
    }
    if (field == 1) {
      *id = b.i();
      return b.ok();
    }
    return false;
  });
}

inline bool parse_tagged(const uint8_t* p, size_t n, int64_t* tag, Bytes* payload) {
  Reader r(p, n), body(nullptr, 0);
  int64_t type = 0;
  if (!r.value(&body, &type)) return false;
  *tag = 0;
  payload->clear();
  return struct_fields(body, [&](int field, Reader& b) {
    if (field == 0) {
      *tag = b.i();
      return b.ok();
    }
    if (field == 1) {
      const uint64_t len = b.u();
      const uint8_t* q = b.ok() ? b.take((size_t)len) : nullptr;
      if (!q) return false;
      payload->assign(q, q + len);
      return true;
    }
    return false;
  });
}

// What a user value turned out to be: the type id of its value message and, for slices, its element kind
// (from the slice definition that precedes it: a fresh encoder always sends it).
struct ValueHead {
  int64_t type = 0;
  int elem = 0;       // kFloat / kInt for a user slice type
  uint64_t count = 0;  // elements (bytes for kByteSlice / kString)
};

// positions `body` at the first element
inline bool open_value(const uint8_t* p, size_t n, Reader* body, ValueHead* h) {
  Reader r(p, n), msg(nullptr, 0);
  while (r.message(&msg)) {
    const int64_t id = msg.i();
    if (!msg.ok()) return false;
    if (id < 0) {  // a definition: remember the element type of a slice
      if (msg.u() == 2) {  // wireType.SliceT
        // sliceType{CommonType{Name, Id}, Elem}
        struct_fields(msg, [&](int field, Reader& b) {
          if (field == 0) return struct_fields(b, [&](int cf, Reader& c) {
              if (cf == 0) {
                const uint64_t len = c.u();
                return c.take((size_t)len) != nullptr;
              }
              c.i();
              return c.ok();
            });
          if (field == 1) {
            h->elem = (int)b.i();
            return b.ok();
          }
          return false;
        });
      }
      continue;
    }
    h->type = id;
    if (msg.u() != 0) return false;  // top-level non-struct value: one zero byte
    h->count = msg.u();
    *body = msg;
    return msg.ok();
  }
  return false;
}

}  // namespace gobwire
}  // namespace mpi
