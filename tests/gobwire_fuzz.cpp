// TEST INFRASTRUCTURE: noise and mutated valid messages into the product's gob reader (mpi_amd/host/gobwire.hpp) -- what mpi::Network
// feeds with whatever a socket delivers before and after the handshake (network.go:242-351, 595-625).  Built with
// -fsanitize=address,undefined by tests/test_tcp_backend.py: no read outside the buffer, no overflow, no crash; "ok <parsed>" at the end.
#include "gobwire.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>
using namespace mpi::gobwire;
int main(int argc, char** argv) {
  std::mt19937_64 rng(12345);
  size_t parsed = 0;
  const int iters = argc > 1 ? atoi(argv[1]) : 100000;
  for (int it = 0; it < iters; it++) {
    Bytes b;
    const int kind = it % 4;
    if (kind == 0) {  // pure noise
      b.resize(rng() % 64);
      for (auto& x : b) x = (uint8_t)rng();
    } else {  // a valid message, mutated
      std::vector<double> v(rng() % 8);
      for (auto& x : v) x = (double)(int64_t)rng() / 7.0;
      Bytes payload = kind == 1 ? value_slice<double>(v.data(), v.size(), "[]float64", kFloat) : value_bytes((const uint8_t*)"hello gob", 9, kind == 2);
      b = kind == 3 ? initial_message("pw", (int64_t)rng()) : tagged_message((int64_t)(rng() % 1000) - 500, payload.data(), payload.size());
      const int flips = (int)(rng() % 4);
      for (int f = 0; f < flips && !b.empty(); f++) b[rng() % b.size()] ^= (uint8_t)(1u << (rng() % 8));
      if (rng() % 5 == 0 && !b.empty()) b.resize(rng() % b.size());
    }
    std::string pw;
    int64_t id = 0, tag = 0;
    Bytes pl;
    parsed += parse_initial(b.data(), b.size(), &pw, &id);
    if (parse_tagged(b.data(), b.size(), &tag, &pl)) {
      parsed++;
      Reader body(nullptr, 0);
      ValueHead h;
      if (open_value(pl.data(), pl.size(), &body, &h)) parsed++;
    }
  }
  printf("ok %zu\n", parsed);
  return 0;
}
