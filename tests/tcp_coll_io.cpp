// one rank of a job on the product's TCP backend (mpi::Network): reads its input vector from a file, runs
// Allreduce / Reduce (root = last rank) / Allgather / Bcast (root 0) on it, writes the results to files -- the test
// (tests/test_tcp_backend.py) compares them with the CPU oracle bit for bit.
//   tcp_coll_io <f32|f64|i64|i32> <sum|prod|min|max> <in.bin> <out-prefix> -mpi-addr A -mpi-alladdr CSV
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "network.hpp"

template <typename T>
static int run(xmpi_op op, const char* in, const std::string& out) {
  std::vector<T> x;
  if (FILE* f = fopen(in, "rb")) {
    fseek(f, 0, SEEK_END);
    x.resize((size_t)ftell(f) / sizeof(T));
    fseek(f, 0, SEEK_SET);
    if (fread(x.data(), sizeof(T), x.size(), f) != x.size()) return 3;
    fclose(f);
  } else {
    return 3;
  }
  const int rank = mpi::Rank(), size = mpi::Size();
  auto dump = [&](const std::vector<T>& v, const char* what) {
    FILE* f = fopen((out + "." + what).c_str(), "wb");
    fwrite(v.data(), sizeof(T), v.size(), f);
    fclose(f);
  };
  std::vector<T> ar, red, ag, bc = x;
  if (mpi::Error e = mpi::Allreduce(mpi::Slice(x), mpi::Into(&ar), op)) return fprintf(stderr, "%s\n", e.What().c_str()), 1;
  if (mpi::Error e = mpi::Reduce(mpi::Slice(x), mpi::Into(&red), op, size - 1)) return fprintf(stderr, "%s\n", e.What().c_str()), 1;
  if (mpi::Error e = mpi::Allgather(mpi::Slice(x), mpi::Into(&ag))) return fprintf(stderr, "%s\n", e.What().c_str()), 1;
  if (mpi::Error e = mpi::Bcast(mpi::Into(&bc), 0)) return fprintf(stderr, "%s\n", e.What().c_str()), 1;
  if (mpi::Error e = mpi::Barrier()) return fprintf(stderr, "%s\n", e.What().c_str()), 1;
  dump(ar, "allreduce");
  if (rank == size - 1) dump(red, "reduce");
  dump(ag, "allgather");
  dump(bc, "bcast");
  return 0;
}

int main(int argc, char** argv) {
  mpi::ParseFlags(&argc, argv);
  if (argc < 5) return 2;
  static mpi::Network net;
  mpi::Register(&net);
  if (mpi::Error e = mpi::Init()) return fprintf(stderr, "%s\n", e.What().c_str()), 1;
  const std::string dt = argv[1], ops = argv[2];
  const xmpi_op op = ops == "sum" ? XMPI_SUM : ops == "prod" ? XMPI_PROD : ops == "min" ? XMPI_MIN : XMPI_MAX;
  int rc = 2;
  if (dt == "f32") rc = run<float>(op, argv[3], argv[4]);
  else if (dt == "f64") rc = run<double>(op, argv[3], argv[4]);
  else if (dt == "i64") rc = run<int64_t>(op, argv[3], argv[4]);
  else if (dt == "i32") rc = run<int32_t>(op, argv[3], argv[4]);
  mpi::Finalize();
  return rc;
}
