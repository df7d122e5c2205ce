// gob_codec.h -- restatement of the subset of Go's encoding/gob wire format the reference uses.
// TEST INFRASTRUCTURE ONLY (CPU baseline + oracle); nothing under mpi_amd/ includes this.
//
// The reference serialises EVERY payload with encoding/gob (Go standard library, version
// unpinned: the reference has no go.mod): call sites network.go:242,258,318,328 (handshake),
// network.go:539 (user value -> bytes), network.go:562 (message{Tag,Bytes} onto the conn),
// network.go:553,609 (decode message), network.go:597 (bytes -> user value), mpi.go:77-91 (Raw).
// No Go toolchain exists in this image, so the codec is restated from gob's published format
// document ("Encoding Details" of package encoding/gob) and pinned by the known-answer vectors
// that document contains (tests/test_gob.py):
//     uint 7 -> 07,  uint 256 -> FE 01 00,  int -129 -> FE 01 01,  float64 17.0 -> FE 31 40,
//     type Point struct{X,Y int}; Point{22,33} -> the 40-byte stream quoted in the document.
// Value encodings (uint / int / float / bytes / string / slices / struct fields) follow the
// document exactly.  The type-descriptor preambles for slice types and for the reference's
// message / initialMessage structs are built by the same rules; their type ids (65, 66, ...)
// depend on Go's process-global allocation order and are NOT verifiable here -- they do not
// affect the payload arithmetic (gob is lossless) nor, measurably, the timing.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace gob {

typedef std::vector<uint8_t> Buf;

// ---- primitives --------------------------------------------------------------------------------
// unsigned: < 128 one byte; else (negated byte count) then big-endian minimal bytes
inline void put_uint(Buf& o, uint64_t x) {
  if (x < 128) {
    o.push_back((uint8_t)x);
    return;
  }
  uint8_t tmp[8];
  int n = 0;
  while (x) {
    tmp[n++] = (uint8_t)(x & 0xFF);
    x >>= 8;
  }
  o.push_back((uint8_t)(256 - n));  // == (uint8_t)(-n)
  while (n) o.push_back(tmp[--n]);
}

// signed: bit 0 = complement flag, value shifted up by one
inline void put_int(Buf& o, int64_t i) {
  uint64_t u = (i < 0) ? ((~(uint64_t)i << 1) | 1) : ((uint64_t)i << 1);
  put_uint(o, u);
}

// float: IEEE-754 float64 bits, byte-reversed (exponent and high mantissa first), as unsigned
inline uint64_t reverse_bytes(uint64_t v) { return __builtin_bswap64(v); }
inline void put_float(Buf& o, double d) {
  uint64_t b;
  memcpy(&b, &d, 8);
  put_uint(o, reverse_bytes(b));
}

struct Reader {
  const uint8_t* p;
  const uint8_t* e;
  bool ok = true;
  Reader(const uint8_t* b, size_t n) : p(b), e(b + n) {}
  uint64_t uint_() {
    if (p >= e) { ok = false; return 0; }
    uint8_t c = *p++;
    if (c < 128) return c;
    int n = 256 - (int)c;
    if (n > 8 || p + n > e) { ok = false; return 0; }
    uint64_t x = 0;
    while (n--) x = (x << 8) | *p++;
    return x;
  }
  int64_t int_() {
    uint64_t u = uint_();
    return (u & 1) ? (int64_t)~(u >> 1) : (int64_t)(u >> 1);
  }
  double float_() {
    uint64_t b = reverse_bytes(uint_());
    double d;
    memcpy(&d, &b, 8);
    return d;
  }
  bool bytes(const uint8_t** out, size_t n) {
    if ((size_t)(e - p) < n) { ok = false; return false; }
    *out = p;
    p += n;
    return true;
  }
};

// predefined type ids (package gob: bootstrap types)
enum { tBool = 1, tInt = 2, tUint = 3, tFloat = 4, tBytes = 5, tString = 6 };
// wireType field numbers
enum { wArrayT = 0, wSliceT = 1, wStructT = 2, wMapT = 3, wGobEncoderT = 4 };
constexpr int kFirstUserId = 65;  // as in the format document's example

inline void put_string(Buf& o, const std::string& s) {
  put_uint(o, s.size());
  o.insert(o.end(), s.begin(), s.end());
}

// a complete gob message = uint(byte length) ++ body
inline void frame(Buf& out, const Buf& body) {
  put_uint(out, body.size());
  out.insert(out.end(), body.begin(), body.end());
}

// CommonType{Name string; Id typeId} as a struct value (both fields non-zero)
inline void put_common(Buf& o, const std::string& name, int id) {
  o.push_back(1);  // field 0: Name
  put_string(o, name);
  o.push_back(1);  // field 1: Id
  put_int(o, id);
  o.push_back(0);
}

// type definition message: (-id) wireType{SliceT: sliceType{CommonType, Elem}}
inline void def_slice(Buf& out, int id, const std::string& name, int elem) {
  Buf b;
  put_int(b, -id);
  b.push_back(wSliceT + 1);  // delta from -1 to field 1
  b.push_back(1);            // sliceType field 0: CommonType
  put_common(b, name, id);
  b.push_back(1);            // field 1: Elem
  put_int(b, elem);
  b.push_back(0);            // end sliceType
  b.push_back(0);            // end wireType
  frame(out, b);
}

struct FieldDef {
  std::string name;
  int id;
};

// (-id) wireType{StructT: structType{CommonType, Field []fieldType{Name,Id}}}
inline void def_struct(Buf& out, int id, const std::string& name, const std::vector<FieldDef>& fields) {
  Buf b;
  put_int(b, -id);
  b.push_back(wStructT + 1);
  b.push_back(1);
  put_common(b, name, id);
  b.push_back(1);  // field 1: Field
  put_uint(b, fields.size());
  for (const FieldDef& f : fields) {
    b.push_back(1);
    put_string(b, f.name);
    b.push_back(1);
    put_int(b, f.id);
    b.push_back(0);
  }
  b.push_back(0);
  b.push_back(0);
  frame(out, b);
}

// (-id) wireType{GobEncoderT: gobEncoderType{CommonType}}
inline void def_gobencoder(Buf& out, int id, const std::string& name) {
  Buf b;
  put_int(b, -id);
  b.push_back(wGobEncoderT + 1);
  b.push_back(1);
  put_common(b, name, id);
  b.push_back(0);
  b.push_back(0);
  frame(out, b);
}

// ---- top-level values as a fresh gob.Encoder would send them (network.go:539) -------------------
// non-struct top-level value: id, a zero "singleton" delta byte, then the value

inline void encode_f64_slice(Buf& out, const double* v, size_t n) {
  def_slice(out, kFirstUserId, "[]float64", tFloat);
  Buf b;
  b.reserve(n * 9 + 16);
  put_int(b, kFirstUserId);
  b.push_back(0);
  put_uint(b, n);
  for (size_t i = 0; i < n; i++) put_float(b, v[i]);
  frame(out, b);
}

inline void encode_f32_slice(Buf& out, const float* v, size_t n) {  // float32 travels widened to float64
  def_slice(out, kFirstUserId, "[]float32", tFloat);
  Buf b;
  b.reserve(n * 6 + 16);
  put_int(b, kFirstUserId);
  b.push_back(0);
  put_uint(b, n);
  for (size_t i = 0; i < n; i++) put_float(b, (double)v[i]);
  frame(out, b);
}

inline void encode_i64_slice(Buf& out, const int64_t* v, size_t n) {
  def_slice(out, kFirstUserId, "[]int64", tInt);
  Buf b;
  b.reserve(n * 9 + 16);
  put_int(b, kFirstUserId);
  b.push_back(0);
  put_uint(b, n);
  for (size_t i = 0; i < n; i++) put_int(b, v[i]);
  frame(out, b);
}

inline void encode_bytes(Buf& out, const uint8_t* v, size_t n) {  // []byte is predefined: no descriptor
  Buf b;
  b.reserve(n + 16);
  put_int(b, tBytes);
  b.push_back(0);
  put_uint(b, n);
  b.insert(b.end(), v, v + n);
  frame(out, b);
}

inline void encode_string(Buf& out, const std::string& s) {
  Buf b;
  put_int(b, tString);
  b.push_back(0);
  put_string(b, s);
  frame(out, b);
}

// Reads one framed message; returns false at end / on error.
inline bool next_message(Reader& r, Reader* body) {
  if (r.p >= r.e) return false;
  uint64_t len = r.uint_();
  const uint8_t* b;
  if (!r.ok || !r.bytes(&b, (size_t)len)) return false;
  *body = Reader(b, (size_t)len);
  return true;
}

// Skips type definitions (negative ids) and positions `body` after the id of the value message.
inline bool value_message(Reader& r, Reader* body, int64_t* id) {
  for (;;) {
    if (!next_message(r, body)) return false;
    *id = body->int_();
    if (!body->ok) return false;
    if (*id >= 0) return true;
  }
}

inline bool decode_f64_slice(const uint8_t* p, size_t n, std::vector<double>* out) {
  Reader r(p, n), b(nullptr, 0);
  int64_t id;
  if (!value_message(r, &b, &id)) return false;
  if (b.uint_() != 0) return false;  // singleton delta
  uint64_t cnt = b.uint_();
  out->resize((size_t)cnt);  // Go re-uses the caller's slice when capacity suffices (bounce.go:89,94)
  for (uint64_t i = 0; i < cnt; i++) (*out)[(size_t)i] = b.float_();
  return b.ok;
}

inline bool decode_f32_slice(const uint8_t* p, size_t n, std::vector<float>* out) {
  Reader r(p, n), b(nullptr, 0);
  int64_t id;
  if (!value_message(r, &b, &id)) return false;
  if (b.uint_() != 0) return false;
  uint64_t cnt = b.uint_();
  out->resize((size_t)cnt);
  for (uint64_t i = 0; i < cnt; i++) (*out)[(size_t)i] = (float)b.float_();  // exact: it was a float32
  return b.ok;
}

inline bool decode_i64_slice(const uint8_t* p, size_t n, std::vector<int64_t>* out) {
  Reader r(p, n), b(nullptr, 0);
  int64_t id;
  if (!value_message(r, &b, &id)) return false;
  if (b.uint_() != 0) return false;
  uint64_t cnt = b.uint_();
  out->resize((size_t)cnt);
  for (uint64_t i = 0; i < cnt; i++) (*out)[(size_t)i] = b.int_();
  return b.ok;
}

inline bool decode_bytes(const uint8_t* p, size_t n, std::vector<uint8_t>* out) {
  Reader r(p, n), b(nullptr, 0);
  int64_t id;
  if (!value_message(r, &b, &id) || id != tBytes) return false;
  if (b.uint_() != 0) return false;
  uint64_t cnt = b.uint_();
  const uint8_t* q;
  if (!b.bytes(&q, (size_t)cnt)) return false;
  out->assign(q, q + cnt);
  return true;
}

inline bool decode_string(const uint8_t* p, size_t n, std::string* out) {
  Reader r(p, n), b(nullptr, 0);
  int64_t id;
  if (!value_message(r, &b, &id) || id != tString) return false;
  if (b.uint_() != 0) return false;
  uint64_t cnt = b.uint_();
  const uint8_t* q;
  if (!b.bytes(&q, (size_t)cnt)) return false;
  out->assign((const char*)q, (size_t)cnt);
  return true;
}

// ---- the reference's wire structs ---------------------------------------------------------------
// type message struct { Tag int; Bytes Raw }   (network.go:511-514; Raw is a GobEncoder, mpi.go:75-91)
// Fresh encoder per message (network.go:562): descriptors are re-sent every time.
inline void encode_message(Buf& out, int64_t tag, const uint8_t* bytes, size_t n) {
  def_struct(out, kFirstUserId, "message", {{"Tag", tInt}, {"Bytes", kFirstUserId + 1}});
  def_gobencoder(out, kFirstUserId + 1, "Raw");
  Buf b;
  b.reserve(n + 32);
  put_int(b, kFirstUserId);
  int last = -1;
  if (tag != 0) {  // zero-valued fields are omitted
    b.push_back((uint8_t)(0 - last));
    put_int(b, tag);
    last = 0;
  }
  if (n != 0) {  // Raw.GobEncode copies the payload (mpi.go:77-81): one more full-size memcpy
    b.push_back((uint8_t)(1 - last));
    put_uint(b, n);
    b.insert(b.end(), bytes, bytes + n);
  }
  b.push_back(0);
  frame(out, b);
}

inline bool decode_message(const uint8_t* p, size_t n, int64_t* tag, std::vector<uint8_t>* bytes, size_t* used) {
  Reader r(p, n), b(nullptr, 0);
  int64_t id;
  if (!value_message(r, &b, &id)) return false;
  *tag = 0;
  bytes->clear();
  int field = -1;
  for (;;) {
    uint64_t delta = b.uint_();
    if (!b.ok) return false;
    if (delta == 0) break;
    field += (int)delta;
    if (field == 0) {
      *tag = b.int_();
    } else if (field == 1) {
      uint64_t cnt = b.uint_();
      const uint8_t* q;
      if (!b.bytes(&q, (size_t)cnt)) return false;
      bytes->assign(q, q + cnt);  // Raw.GobDecode copies (mpi.go:83-91)
    } else {
      return false;
    }
  }
  if (used) *used = (size_t)(r.p - p);
  return b.ok;
}

// type initialMessage struct { Password string; Id int }   (network.go:198-201)
inline void encode_initial(Buf& out, const std::string& password, int64_t idv) {
  def_struct(out, kFirstUserId, "initialMessage", {{"Password", tString}, {"Id", tInt}});
  Buf b;
  put_int(b, kFirstUserId);
  int last = -1;
  if (!password.empty()) {
    b.push_back((uint8_t)(0 - last));
    put_string(b, password);
    last = 0;
  }
  if (idv != 0) {
    b.push_back((uint8_t)(1 - last));
    put_int(b, idv);
  }
  b.push_back(0);
  frame(out, b);
}

inline bool decode_initial(const uint8_t* p, size_t n, std::string* password, int64_t* idv, size_t* used) {
  Reader r(p, n), b(nullptr, 0);
  int64_t id;
  if (!value_message(r, &b, &id)) return false;
  password->clear();
  *idv = 0;
  int field = -1;
  for (;;) {
    uint64_t delta = b.uint_();
    if (!b.ok) return false;
    if (delta == 0) break;
    field += (int)delta;
    if (field == 0) {
      uint64_t cnt = b.uint_();
      const uint8_t* q;
      if (!b.bytes(&q, (size_t)cnt)) return false;
      password->assign((const char*)q, (size_t)cnt);
    } else if (field == 1) {
      *idv = b.int_();
    } else {
      return false;
    }
  }
  if (used) *used = (size_t)(r.p - p);
  return b.ok;
}

}  // namespace gob
