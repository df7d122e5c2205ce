// devprobe -- what the box shows: devices, their XCC / CU counts, the peer-access matrix, and (with >= 2 devices) a
// peer copy each way, by the copy engine and by a kernel that reads / writes the peer's memory directly.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/devprobe_bin scripts/devprobe.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      printf("{\"error\": \"%s -> %s\"}\n", #x, hipGetErrorString(e_));                        \
      return 1;                                                                                \
    }                                                                                          \
  } while (0)

__global__ void pull16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  int n = 0;
  CK(hipGetDeviceCount(&n));
  printf("{\"devices\": %d, \"props\": [", n);
  for (int d = 0; d < n; d++) {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, d));
    size_t fr = 0, tot = 0;
    CK(hipSetDevice(d));
    CK(hipMemGetInfo(&fr, &tot));
    printf("%s{\"id\": %d, \"name\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"mem_GiB\": %.1f, \"free_GiB\": %.1f, \"pci\": \"%04x:%02x:%02x\"}",
           d ? ", " : "", d, p.name, p.gcnArchName, p.multiProcessorCount, tot / 1073741824.0, fr / 1073741824.0, p.pciDomainID,
           p.pciBusID, p.pciDeviceID);
  }
  printf("], \"peer\": [");
  for (int a = 0; a < n; a++) {
    printf("%s[", a ? ", " : "");
    for (int b = 0; b < n; b++) {
      int can = a == b;
      if (a != b) CK(hipDeviceCanAccessPeer(&can, a, b));
      printf("%s%d", b ? ", " : "", can);
    }
    printf("]");
  }
  printf("]");
  if (n >= 2) {
    const size_t bytes = argc > 1 ? strtoull(argv[1], nullptr, 0) : (256u << 20);
    void *a = nullptr, *b = nullptr;
    CK(hipSetDevice(1));
    CK(hipMalloc(&b, bytes));
    CK(hipMemset(b, 1, bytes));
    CK(hipDeviceSynchronize());
    CK(hipSetDevice(0));
    hipError_t pe = hipDeviceEnablePeerAccess(1, 0);
    printf(", \"enable_peer_0_1\": \"%s\"", hipGetErrorString(pe));
    (void)hipGetLastError();
    CK(hipMalloc(&a, bytes));
    CK(hipMemset(a, 2, bytes));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    CK(hipDeviceSynchronize());
    double best[4] = {1e9, 1e9, 1e9, 1e9};
    for (int it = 0; it < 6; it++) {
      double t0 = now();
      CK(hipMemcpyPeerAsync(a, 0, b, 1, bytes, s));
      CK(hipStreamSynchronize(s));
      double t1 = now();
      CK(hipMemcpyPeerAsync(b, 1, a, 0, bytes, s));
      CK(hipStreamSynchronize(s));
      double t2 = now();
      if (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) {
        pull16<<<2048, 256, 0, s>>>((const uint4*)b, (uint4*)a, bytes / 16);  // kernel on device 0 READS device 1
        CK(hipStreamSynchronize(s));
        double t3 = now();
        pull16<<<2048, 256, 0, s>>>((const uint4*)a, (uint4*)b, bytes / 16);  // kernel on device 0 WRITES device 1
        CK(hipStreamSynchronize(s));
        double t4 = now();
        if (t3 - t2 < best[2]) best[2] = t3 - t2;
        if (t4 - t3 < best[3]) best[3] = t4 - t3;
      }
      if (t1 - t0 < best[0]) best[0] = t1 - t0;
      if (t2 - t1 < best[1]) best[1] = t2 - t1;
    }
    printf(", \"bytes\": %zu, \"GBps\": {\"memcpy_peer_1to0\": %.1f, \"memcpy_peer_0to1\": %.1f, \"kernel_reads_peer\": %.1f, \"kernel_writes_peer\": %.1f}",
           bytes, bytes / best[0] / 1e9, bytes / best[1] / 1e9, best[2] < 1e8 ? bytes / best[2] / 1e9 : 0.0,
           best[3] < 1e8 ? bytes / best[3] / 1e9 : 0.0);
  }
  printf("}\n");
  return 0;
}
