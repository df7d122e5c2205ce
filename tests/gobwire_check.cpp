// known answers of gob's format document ("Encoding Details" of package encoding/gob) for the product-side
// codec (mpi_amd/host/gobwire.hpp), plus round trips of the reference's two wire structs and of user values
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "gobwire.hpp"

using namespace mpi::gobwire;

static std::string hex(const Bytes& b) {
  std::string s;
  char t[4];
  for (uint8_t c : b) {
    snprintf(t, sizeof t, "%02X", c);
    s += t;
  }
  return s;
}

int main() {
  int bad = 0;
  auto expect = [&](const std::string& got, const char* want, const char* what) {
    if (got != want) {
      printf("%s: got %s want %s\n", what, got.c_str(), want);
      bad++;
    }
  };
  { Writer w; w.u(7); expect(hex(w.out), "07", "uint 7"); }
  { Writer w; w.u(256); expect(hex(w.out), "FE0100", "uint 256"); }
  { Writer w; w.i(-129); expect(hex(w.out), "FE0101", "int -129"); }
  { Writer w; w.f(17.0); expect(hex(w.out), "FE3140", "float 17.0"); }
  {  // type Point struct{X, Y int}; Point{22, 33}: the document's 40-byte stream
    static const Field f[] = {{"X", kInt}, {"Y", kInt}};
    Bytes s;
    define_struct(&s, 65, "Point", f, 2);
    Writer w;
    w.i(65);
    w.u(1); w.i(22);
    w.u(1); w.i(33);
    w.u(0);
    put_message(&s, w);
    expect(hex(s), "1FFF810301010550" "6F696E7401FF8200" "0102010158010400" "0101590104000000" "07FF82012C014200", "Point{22,33}");
  }
  {  // the reference's wire structs survive a round trip, zero fields omitted
    std::string pw;
    int64_t id = -1;
    Bytes m = initial_message("secret", 3);
    if (!parse_initial(m.data(), m.size(), &pw, &id) || pw != "secret" || id != 3) bad++, printf("initialMessage round trip\n");
    m = initial_message("", 0);
    if (!parse_initial(m.data(), m.size(), &pw, &id) || !pw.empty() || id != 0) bad++, printf("initialMessage zero values\n");
    const uint8_t payload[5] = {1, 2, 3, 250, 0};
    int64_t tag = -1;
    Bytes got;
    m = tagged_message(-7, payload, 5);
    if (!parse_tagged(m.data(), m.size(), &tag, &got) || tag != -7 || got != Bytes(payload, payload + 5)) bad++, printf("message round trip\n");
    m = tagged_message(0, nullptr, 0);  // the ack of tag 0: every field zero
    if (!parse_tagged(m.data(), m.size(), &tag, &got) || tag != 0 || !got.empty()) bad++, printf("ack round trip\n");
  }
  {  // user values
    const double v[4] = {0.0, -1.5, 3.141592653589793, 1e-300};
    Bytes m = value_slice(v, 4, "[]float64", kFloat);
    Reader body(nullptr, 0);
    ValueHead h;
    if (!open_value(m.data(), m.size(), &body, &h) || h.type != 65 || h.elem != kFloat || h.count != 4) bad++, printf("[]float64 head\n");
    for (int k = 0; k < 4; k++)
      if (body.f() != v[k]) bad++, printf("[]float64[%d]\n", k);
    const int64_t iv[3] = {0, -9223372036854775807LL - 1, 9223372036854775807LL};
    m = value_slice(iv, 3, "[]int64", kInt);
    ValueHead hi;
    if (!open_value(m.data(), m.size(), &body, &hi) || hi.elem != kInt || hi.count != 3) bad++, printf("[]int64 head\n");
    for (int k = 0; k < 3; k++)
      if (body.i() != iv[k]) bad++, printf("[]int64[%d]\n", k);
    m = value_bytes((const uint8_t*)"hello", 5, true);
    ValueHead hs;
    if (!open_value(m.data(), m.size(), &body, &hs) || hs.type != kString || hs.count != 5) bad++, printf("string head\n");
  }
  {  // every primitive survives a round trip, for a few thousand pseudo-random values of every magnitude
    uint64_t s = 88172645463325252ull;
    auto rnd = [&s]() {
      s ^= s << 13;
      s ^= s >> 7;
      s ^= s << 17;
      return s;
    };
    for (int k = 0; k < 5000 && !bad; k++) {
      const uint64_t u = rnd() >> (rnd() % 64);
      const int64_t i = (int64_t)(rnd() >> (rnd() % 64)) * ((rnd() & 1) ? 1 : -1);
      uint64_t bits = rnd();
      if (((bits >> 52) & 0x7FF) == 0x7FF) bits &= ~(1ull << 62);  // no NaN / inf: compared by value below
      double d;
      memcpy(&d, &bits, 8);
      Writer w;
      w.u(u);
      w.i(i);
      w.f(d);
      w.f((double)(float)d);
      Reader r(w.out.data(), w.out.size());
      if (r.u() != u || r.i() != i || r.f() != d || r.f() != (double)(float)d || !r.ok() || !r.at_end()) bad++, printf("primitive round trip %d\n", k);
    }
    { Writer w; w.i(INT64_MIN); w.i(INT64_MAX); w.u(UINT64_MAX); Reader r(w.out.data(), w.out.size());
      if (r.i() != INT64_MIN || r.i() != INT64_MAX || r.u() != UINT64_MAX) bad++, printf("extremes\n"); }
  }
  printf(bad ? "FAILED\n" : "ok\n");
  return bad ? 1 : 0;
}
