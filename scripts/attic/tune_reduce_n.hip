// tune_reduce_n.hip -- A/B harness for the N-way rank-order fold (development tool, not product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/tune_reduce_n.hip -o scripts/tune_reduce_n_bin
// Variants: packets per lane (U), non-temporal loads, block size, sources walked per lane vs per wave.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x)                                                \
  do {                                                       \
    hipError_t e = (x);                                      \
    if (e != hipSuccess) {                                   \
      printf("%s: %s\n", #x, hipGetErrorString(e));          \
      exit(1);                                               \
    }                                                        \
  } while (0)

struct Ptrs {
  const f4* p[8];
};

template <int NT>
__device__ __forceinline__ f4 ld(const f4* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}

// U packets per lane, all 8*U loads issued before the first add
template <int BLOCK, int U, int NT>
__global__ __launch_bounds__(BLOCK) void k_fold(f4* dst, Ptrs s, size_t npack) {
  constexpr size_t kTile = (size_t)BLOCK * U;
  const size_t stride = (size_t)gridDim.x * kTile;
  const size_t lane_off = (size_t)(threadIdx.x >> 6) * (64 * U) + (threadIdx.x & 63);
  for (size_t base = (size_t)blockIdx.x * kTile; base + kTile <= npack; base += stride) {
    const size_t first = base + lane_off;
    f4 v[8][U];
#pragma unroll
    for (int q = 0; q < 8; q++)
#pragma unroll
      for (int k = 0; k < U; k++) v[q][k] = ld<NT>(s.p[q] + first + k * 64);
#pragma unroll
    for (int k = 0; k < U; k++) {
      f4 acc = v[0][k];
#pragma unroll
      for (int q = 1; q < 8; q++) acc = acc + v[q][k];
      dst[first + k * 64] = acc;
    }
  }
}

template <int BLOCK, int U, int NT>
void run(f4* d, Ptrs s, size_t npack, int gridcap, int reps) {
  size_t tiles = npack / ((size_t)BLOCK * U);
  int grid = (int)(gridcap > 0 && tiles > (size_t)gridcap ? gridcap : tiles);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_fold<BLOCK, U, NT>), dim3(grid), dim3(BLOCK), 0, 0, d, s, npack);
  CK(hipDeviceSynchronize());
  double tot = 0, best = 1e30;
  for (int i = 0; i < reps; i++) {
    hipExtLaunchKernelGGL((k_fold<BLOCK, U, NT>), dim3(grid), dim3(BLOCK), 0, 0, e0, e1, 0, d, s, npack);
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    tot += ms;
    best = ms < best ? ms : best;
  }
  const double bytes = 9.0 * 16.0 * (double)npack;
  printf("  block %4d U %d nt %d gridcap %5d grid %6d : mean %8.2f us %7.1f GB/s  best %8.2f us %7.1f GB/s\n", BLOCK, U, NT, gridcap, grid,
         1e3 * tot / reps, bytes / (tot / reps * 1e-3) / 1e9, 1e3 * best, bytes / (best * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  setvbuf(stdout, nullptr, _IONBF, 0);
  for (size_t mib : {32, 128}) {
    const size_t bytes = mib << 20, npack = bytes / 16;
    Ptrs s;
    f4* d;
    // sources rotate through 3 sets so the operands are cold for the small size as well
    f4* all[24];
    for (int i = 0; i < 24; i++) {
      CK(hipMalloc(&all[i], bytes));
      CK(hipMemset(all[i], i + 1, bytes));
    }
    CK(hipMalloc(&d, bytes));
    printf("== 8 sources of %zu MiB (traffic %zu MiB)\n", mib, 9 * mib);
    static int rot = 0;
    auto pick = [&]() {
      for (int q = 0; q < 8; q++) s.p[q] = all[(rot * 8 + q) % 24];
      rot++;
    };
    pick(); run<256, 1, 0>(d, s, npack, 0, reps);
    pick(); run<256, 1, 1>(d, s, npack, 0, reps);
    pick(); run<256, 1, 1>(d, s, npack, 2048, reps);
    pick(); run<256, 2, 0>(d, s, npack, 0, reps);
    pick(); run<256, 2, 1>(d, s, npack, 0, reps);
    pick(); run<256, 4, 1>(d, s, npack, 0, reps);
    pick(); run<512, 1, 1>(d, s, npack, 0, reps);
    pick(); run<512, 2, 1>(d, s, npack, 0, reps);
    pick(); run<1024, 1, 1>(d, s, npack, 0, reps);
    pick(); run<128, 2, 1>(d, s, npack, 0, reps);
    pick(); run<64, 4, 1>(d, s, npack, 0, reps);
    for (int i = 0; i < 24; i++) CK(hipFree(all[i]));
    CK(hipFree(d));
  }
  return 0;
}
