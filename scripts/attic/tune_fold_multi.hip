// tune_fold_multi.hip -- A/B harness for the zero-copy allreduce kernel (development tool, not product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/tune_fold_multi.hip -o scripts/tune_fold_multi_bin
// Emulates the collective on one GPU: R "ranks", each with its own send and receive buffer of S MiB;
// one step = R launches back to back, launch j folds chunk j of the R send buffers (rank order) and
// stores it into chunk j of the R receive buffers.  Reports the step time and the HBM rate
// (2 * R * S bytes per step).  Variants: packets per lane (U), block size, nt loads / stores, grid cap.
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

constexpr int R = 8;
struct Ptrs {
  const f4* s[R];
  f4* d[R];
};

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

// U packets per lane per source, all R*U loads issued before the first add; wave-contiguous U KiB
template <int BLOCK, int U, int NTL, int NTS>
__global__ __launch_bounds__(BLOCK) void k_fold(Ptrs q, size_t npack) {
  constexpr size_t kTile = (size_t)BLOCK * U;
  const size_t stride = (size_t)gridDim.x * kTile;
  const size_t lane_off = (size_t)(threadIdx.x >> 6) * (64 * U) + (threadIdx.x & 63);
  for (size_t base = (size_t)blockIdx.x * kTile; base + kTile <= npack; base += stride) {
    const size_t first = base + lane_off;
    f4 v[R][U];
#pragma unroll
    for (int s = 0; s < R; s++)
#pragma unroll
      for (int k = 0; k < U; k++) v[s][k] = ld<NTL>(q.s[s] + first + k * 64);
#pragma unroll
    for (int k = 0; k < U; k++) {
#pragma unroll
      for (int s = 1; s < R; s++) v[0][k] = v[0][k] + v[s][k];
    }
#pragma unroll
    for (int d = 0; d < R; d++)
#pragma unroll
      for (int k = 0; k < U; k++) st<NTS>(q.d[d] + first + k * 64, v[0][k]);
  }
}

// XCD-contiguous mapping: workgroups are dealt round-robin to the 8 XCDs, so block b runs on XCD b % 8;
// give every XCD one contiguous eighth of the buffer instead of every 8th tile
template <int BLOCK, int U, int NTL, int NTS>
__global__ __launch_bounds__(BLOCK) void k_fold_xcd(Ptrs q, size_t npack) {
  constexpr size_t kTile = (size_t)BLOCK * U;
  const size_t lane_off = (size_t)(threadIdx.x >> 6) * (64 * U) + (threadIdx.x & 63);
  const size_t per = gridDim.x / 8;
  const size_t tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
  const size_t base = tile * kTile;
  if (blockIdx.x >= per * 8 || base + kTile > npack) return;
  const size_t first = base + lane_off;
  f4 v[R][U];
#pragma unroll
  for (int s = 0; s < R; s++)
#pragma unroll
    for (int k = 0; k < U; k++) v[s][k] = ld<NTL>(q.s[s] + first + k * 64);
#pragma unroll
  for (int k = 0; k < U; k++) {
#pragma unroll
    for (int s = 1; s < R; s++) v[0][k] = v[0][k] + v[s][k];
  }
#pragma unroll
  for (int d = 0; d < R; d++)
#pragma unroll
    for (int k = 0; k < U; k++) st<NTS>(q.d[d] + first + k * 64, v[0][k]);
}

// two halves of the sources at a time (half the registers of the full prefetch at the same U)
template <int BLOCK, int U, int NTL, int NTS>
__global__ __launch_bounds__(BLOCK) void k_fold_split(Ptrs q, size_t npack) {
  constexpr size_t kTile = (size_t)BLOCK * U;
  const size_t stride = (size_t)gridDim.x * kTile;
  const size_t lane_off = (size_t)(threadIdx.x >> 6) * (64 * U) + (threadIdx.x & 63);
  for (size_t base = (size_t)blockIdx.x * kTile; base + kTile <= npack; base += stride) {
    const size_t first = base + lane_off;
    f4 a[R / 2][U], b[R / 2][U];
#pragma unroll
    for (int s = 0; s < R / 2; s++)
#pragma unroll
      for (int k = 0; k < U; k++) a[s][k] = ld<NTL>(q.s[s] + first + k * 64);
#pragma unroll
    for (int s = 0; s < R / 2; s++)
#pragma unroll
      for (int k = 0; k < U; k++) b[s][k] = ld<NTL>(q.s[R / 2 + s] + first + k * 64);
#pragma unroll
    for (int k = 0; k < U; k++) {
#pragma unroll
      for (int s = 1; s < R / 2; s++) a[0][k] = a[0][k] + a[s][k];
#pragma unroll
      for (int s = 0; s < R / 2; s++) a[0][k] = a[0][k] + b[s][k];
    }
#pragma unroll
    for (int d = 0; d < R; d++)
#pragma unroll
      for (int k = 0; k < U; k++) st<NTS>(q.d[d] + first + k * 64, a[0][k]);
  }
}

f4* g_send[R];
f4* g_recv[R];
size_t g_chunk_pack;

template <typename K>
void run(const char* name, K kern, int block, int U, int gridcap, int reps) {
  const size_t tiles = g_chunk_pack / ((size_t)block * U);
  const int grid = (int)(gridcap > 0 && tiles > (size_t)gridcap ? gridcap : tiles);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  double tot = 0, best = 1e30;
  for (int i = 0; i < reps + 2; i++) {
    CK(hipEventRecord(e0, st));
    for (int j = 0; j < R; j++) {
      Ptrs q;
      for (int p = 0; p < R; p++) {
        q.s[p] = g_send[p] + (size_t)j * g_chunk_pack;
        q.d[p] = g_recv[(j + p) % R] + (size_t)j * g_chunk_pack;
      }
      hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, st, q, g_chunk_pack);
    }
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (i >= 2) {
      tot += ms;
      best = ms < best ? ms : best;
    }
  }
  const double bytes = 2.0 * R * R * 16.0 * (double)g_chunk_pack;
  printf("  %-28s block %4d U %d gridcap %5d grid %6d : step mean %8.1f us %7.1f GB/s  best %8.1f us %7.1f GB/s\n", name,
         block, U, gridcap, grid, 1e3 * tot / reps, bytes / (tot / reps * 1e-3) / 1e9, 1e3 * best,
         bytes / (best * 1e-3) / 1e9);
  CK(hipStreamDestroy(st));
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  setvbuf(stdout, nullptr, _IONBF, 0);
  for (size_t mib : {256, 32}) {
    const size_t bytes = mib << 20;
    g_chunk_pack = bytes / 16 / R;
    for (int i = 0; i < R; i++) {
      CK(hipMalloc(&g_send[i], bytes));
      CK(hipMalloc(&g_recv[i], bytes));
      CK(hipMemset(g_send[i], i + 1, bytes));
    }
    printf("== %d ranks x %zu MiB (HBM traffic per step %zu MiB)\n", R, mib, 2 * R * mib);
    run("U1 ntl", k_fold<256, 1, 1, 0>, 256, 1, 0, reps);
    run("U1 plain", k_fold<256, 1, 0, 0>, 256, 1, 0, reps);
    run("U1 ntl nts", k_fold<256, 1, 1, 1>, 256, 1, 0, reps);
    run("U1 ntl nts xcd-contig", k_fold_xcd<256, 1, 1, 1>, 256, 1, 0, reps);
    run("U4 ntl nts xcd-contig", k_fold_xcd<256, 4, 1, 1>, 256, 4, 0, reps);
    run("U2 ntl", k_fold<256, 2, 1, 0>, 256, 2, 0, reps);
    run("U2 ntl nts", k_fold<256, 2, 1, 1>, 256, 2, 0, reps);
    run("U4 ntl", k_fold<256, 4, 1, 0>, 256, 4, 0, reps);
    run("U2 split ntl", k_fold_split<256, 2, 1, 0>, 256, 2, 0, reps);
    run("U4 split ntl", k_fold_split<256, 4, 1, 0>, 256, 4, 0, reps);
    run("U4 split ntl nts", k_fold_split<256, 4, 1, 1>, 256, 4, 0, reps);
    run("U1 ntl b512", k_fold<512, 1, 1, 0>, 512, 1, 0, reps);
    run("U1 ntl b1024", k_fold<1024, 1, 1, 0>, 1024, 1, 0, reps);
    run("U1 ntl b128", k_fold<128, 1, 1, 0>, 128, 1, 0, reps);
    run("U1 ntl cap2048", k_fold<256, 1, 1, 0>, 256, 1, 2048, reps);
    run("U2 ntl cap2048", k_fold<256, 2, 1, 0>, 256, 2, 2048, reps);
    run("U1 ntl cap4096", k_fold<256, 1, 1, 0>, 256, 1, 4096, reps);
    for (int i = 0; i < R; i++) {
      CK(hipFree(g_send[i]));
      CK(hipFree(g_recv[i]));
    }
  }
  return 0;
}
