// Reads "kind<TAB>a<TAB>b<TAB>raw hex<TAB>gob hex" lines (tests/test_reference_golden.py writes them from the fixtures the
// REAL reference produced: go/golden/gen_golden.go) and checks that the product-side codec (mpi_amd/host/gobwire.hpp)
// produces the same bytes as Go's encoding/gob did.  kinds: []byte string []float64 []float32 []int64 []int32
// (a, b unused), initialMessage (a = id, b = password), message / ack (a = tag, raw = payload).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "gobwire.hpp"

using namespace mpi::gobwire;

static Bytes unhex(const std::string& s) {
  Bytes b;
  for (size_t i = 0; i + 1 < s.size(); i += 2) b.push_back((uint8_t)strtoul(s.substr(i, 2).c_str(), nullptr, 16));
  return b;
}

template <typename T>
static std::vector<T> as(const Bytes& raw) {
  std::vector<T> v(raw.size() / sizeof(T));
  if (!v.empty()) memcpy(v.data(), raw.data(), v.size() * sizeof(T));
  return v;
}

int main() {
  std::string line;
  int bad = 0, n = 0;
  while (std::getline(std::cin, line)) {
    std::vector<std::string> f;
    std::stringstream ss(line);
    std::string cell;
    while (std::getline(ss, cell, '\t')) f.push_back(cell);
    if (f.size() < 5) f.resize(5);
    const Bytes raw = unhex(f[3]), want = unhex(f[4]);
    Bytes got;
    const std::string& kind = f[0];
    if (kind == "[]byte") got = value_bytes(raw.data(), raw.size(), false);
    else if (kind == "string") got = value_bytes(raw.data(), raw.size(), true);
    else if (kind == "[]float64") { auto v = as<double>(raw); got = value_slice(v.data(), v.size(), "[]float64", kFloat); }
    else if (kind == "[]float32") { auto v = as<float>(raw); got = value_slice(v.data(), v.size(), "[]float32", kFloat); }
    else if (kind == "[]int64") { auto v = as<int64_t>(raw); got = value_slice(v.data(), v.size(), "[]int64", kInt); }
    else if (kind == "[]int32") { auto v = as<int32_t>(raw); got = value_slice(v.data(), v.size(), "[]int32", kInt); }
    else if (kind == "initialMessage") got = initial_message(f[2], atoll(f[1].c_str()));
    else if (kind == "message" || kind == "ack") got = tagged_message(atoll(f[1].c_str()), raw.data(), raw.size());
    else continue;
    n++;
    if (got != want) {
      bad++;
      printf("%s %s %s: %zu bytes here, %zu from Go\n", kind.c_str(), f[1].c_str(), f[2].c_str(), got.size(), want.size());
    }
  }
  printf(bad ? "FAILED %d of %d\n" : "ok %d %d\n", bad, n);
  return bad ? 1 : 0;
}
