// tune_reduce.hip -- A/B harness for the streaming reduction kernel (development tool, not product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/tune_reduce.hip -o /tmp/tune_reduce && /tmp/tune_reduce
// Variants: packets per lane (unroll), non-temporal loads / stores, block size, grid cap.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int pack_t __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                          \
  do {                                                                 \
    hipError_t e = (x);                                                \
    if (e != hipSuccess) {                                             \
      printf("%s failed: %s\n", #x, hipGetErrorString(e));             \
      exit(1);                                                         \
    }                                                                  \
  } while (0)

template <int NT>
__device__ __forceinline__ f4 ld(const f4* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <int NT>
__device__ __forceinline__ void st(f4* p, f4 v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

template <int BLOCK, int UNROLL, int NTL, int NTS>
__global__ __launch_bounds__(BLOCK) void k_reduce(f4* dst, const f4* a, const f4* b, size_t npack) {
  constexpr size_t kTile = (size_t)BLOCK * UNROLL;
  const size_t stride = (size_t)gridDim.x * kTile;
  const size_t lane_off = (size_t)(threadIdx.x >> 6) * (64 * UNROLL) + (threadIdx.x & 63);
  for (size_t base = (size_t)blockIdx.x * kTile; base + kTile <= npack; base += stride) {
    const size_t first = base + lane_off;
    f4 va[UNROLL], vb[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; k++) {
      va[k] = ld<NTL>(a + first + k * 64);
      vb[k] = ld<NTL>(b + first + k * 64);
    }
#pragma unroll
    for (int k = 0; k < UNROLL; k++) st<NTS>(dst + first + k * 64, va[k] + vb[k]);
  }
}

template <int BLOCK, int UNROLL, int NTL, int NTS>
double run(f4* d, const f4* a, const f4* b, size_t npack, int gridcap, int reps) {
  size_t tiles = npack / ((size_t)BLOCK * UNROLL);
  int grid = (int)(gridcap > 0 && tiles > (size_t)gridcap ? gridcap : tiles);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_reduce<BLOCK, UNROLL, NTL, NTS>), dim3(grid), dim3(BLOCK), 0, 0, d, a, b, npack);
  CK(hipDeviceSynchronize());
  double best = 1e30, tot = 0;
  for (int i = 0; i < reps; i++) {
    hipExtLaunchKernelGGL((k_reduce<BLOCK, UNROLL, NTL, NTS>), dim3(grid), dim3(BLOCK), 0, 0, e0, e1, 0, d, a, b, npack);
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
    tot += ms;
  }
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  const double bytes = 3.0 * 16.0 * (double)npack;
  printf("  block %4d unroll %d ntl %d nts %d gridcap %6d grid %6d : mean %8.2f us  %7.1f GB/s   best %8.2f us %7.1f GB/s\n", BLOCK,
         UNROLL, NTL, NTS, gridcap, grid, 1e3 * tot / reps, bytes / (tot / reps * 1e-3) / 1e9, 1e3 * best, bytes / (best * 1e-3) / 1e9);
  return tot / reps;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  for (size_t mib : {32, 256}) {
    const size_t bytes = mib << 20, npack = bytes / 16;
    f4 *a, *b, *d;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc(&d, bytes));
    CK(hipMemset(a, 1, bytes));
    CK(hipMemset(b, 2, bytes));
    printf("== operands of %zu MiB (traffic %zu MiB)\n", mib, 3 * mib);
    for (int cap : {2048, 4096, 0}) {
      run<256, 4, 0, 0>(d, a, b, npack, cap, reps);
      run<256, 4, 1, 0>(d, a, b, npack, cap, reps);
      run<256, 4, 0, 1>(d, a, b, npack, cap, reps);
      run<256, 4, 1, 1>(d, a, b, npack, cap, reps);
    }
    run<256, 2, 0, 0>(d, a, b, npack, 0, reps);
    run<256, 2, 1, 1>(d, a, b, npack, 0, reps);
    run<256, 8, 0, 0>(d, a, b, npack, 2048, reps);
    run<256, 8, 1, 1>(d, a, b, npack, 2048, reps);
    run<256, 8, 1, 1>(d, a, b, npack, 0, reps);
    run<512, 4, 0, 0>(d, a, b, npack, 2048, reps);
    run<512, 4, 1, 1>(d, a, b, npack, 0, reps);
    run<1024, 2, 1, 1>(d, a, b, npack, 0, reps);
    run<1024, 4, 1, 1>(d, a, b, npack, 1024, reps);
    run<128, 4, 1, 1>(d, a, b, npack, 0, reps);
    run<64, 8, 1, 1>(d, a, b, npack, 0, reps);
    CK(hipFree(a));
    CK(hipFree(b));
    CK(hipFree(d));
  }
  return 0;
}
