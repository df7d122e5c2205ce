// tests/devsim -- a HIP runtime for a machine without a GPU (TEST INFRASTRUCTURE; nothing under mpi_amd/ knows about it).
//
// What it is for: the library's host sources AND its gfx950 kernel sources are compiled for the CPU against THIS header
// instead of ROCm's (tests/devsim/build.py: clang++ -x c++ -DXMPI_DEVSIM -I tests/devsim/include), so that
//   * the code path with more than one HIP device -- rank i on device i, hipDeviceEnablePeerAccess, hipIpc handles opened
//     by another process on another device -- executes at all (the GPU boxes of this project have one device), and
//   * the flag protocols of the kernels THAT SHIP (kdev.h dsync_begin / dsync_end, ll.hip, sched.hip) run under
//     ThreadSanitizer with the ranks as threads, instead of Python models of them.
// It is not a product path: libxmpi.so is built by hipcc for gfx950 and fails loudly without a GPU (xmpi_init: "no HIP
// device is visible; xmpi has no CPU fallback"); nothing here is built by mpi_amd.build or loaded by mpi_amd.xmpi.
//
// The subset of the HIP API the library uses, with the semantics it relies on: in-order streams (a worker thread each),
// events, stream capture into replayable graphs, device / pinned / registered memory, IPC handles (device memory is backed
// by POSIX shared memory so that another PROCESS can map it), N virtual devices (DEVSIM_DEVICES).  Kernels: hip_runtime.h.
#pragma once
#include <cstddef>
#include <cstdint>

typedef enum hipError_t {
  hipSuccess = 0,
  hipErrorInvalidValue = 1,
  hipErrorOutOfMemory = 2,
  hipErrorNotInitialized = 3,
  hipErrorInvalidDevicePointer = 17,
  hipErrorInvalidDevice = 101,
  hipErrorInvalidContext = 201,
  hipErrorInvalidHandle = 400,
  hipErrorNotReady = 600,
  hipErrorPeerAccessAlreadyEnabled = 704,
  hipErrorPeerAccessNotEnabled = 705,
  hipErrorHostMemoryAlreadyRegistered = 712,
  hipErrorHostMemoryNotRegistered = 713,
  hipErrorStreamCaptureUnsupported = 900,
  hipErrorStreamCaptureInvalidated = 901,
  hipErrorUnknown = 999,
} hipError_t;

struct ihipStream_t;
struct ihipEvent_t;
struct ihipGraph;
struct hipGraphExec;
struct hipGraphNode;
typedef ihipStream_t* hipStream_t;
typedef ihipEvent_t* hipEvent_t;
typedef ihipGraph* hipGraph_t;
typedef hipGraphExec* hipGraphExec_t;
typedef hipGraphNode* hipGraphNode_t;
typedef void* hipDeviceptr_t;

typedef struct hipIpcMemHandle_st {
  char reserved[64];
} hipIpcMemHandle_t;

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef enum hipMemcpyKind {
  hipMemcpyHostToHost = 0,
  hipMemcpyHostToDevice = 1,
  hipMemcpyDeviceToHost = 2,
  hipMemcpyDeviceToDevice = 3,
  hipMemcpyDefault = 4,
} hipMemcpyKind;

typedef enum hipMemoryType {
  hipMemoryTypeUnregistered = 0,
  hipMemoryTypeHost = 1,
  hipMemoryTypeDevice = 2,
  hipMemoryTypeManaged = 3,
  hipMemoryTypeArray = 10,
  hipMemoryTypeUnified = 11,
} hipMemoryType;

typedef struct hipPointerAttribute_t {
  hipMemoryType type;
  int device;
  void* devicePointer;
  void* hostPointer;
  int isManaged;
  unsigned allocationFlags;
} hipPointerAttribute_t;

typedef enum hipStreamCaptureMode {
  hipStreamCaptureModeGlobal = 0,
  hipStreamCaptureModeThreadLocal = 1,
  hipStreamCaptureModeRelaxed = 2,
} hipStreamCaptureMode;
typedef enum hipStreamCaptureStatus {
  hipStreamCaptureStatusNone = 0,
  hipStreamCaptureStatusActive = 1,
  hipStreamCaptureStatusInvalidated = 2,
} hipStreamCaptureStatus;

#define hipStreamDefault 0x0
#define hipStreamNonBlocking 0x1
#define hipEventDefault 0x0
#define hipEventBlockingSync 0x1
#define hipEventDisableTiming 0x2
#define hipEventInterprocess 0x4
#define hipHostMallocDefault 0x0
#define hipHostMallocPortable 0x1
#define hipHostMallocMapped 0x2
#define hipHostRegisterDefault 0x0
#define hipHostRegisterPortable 0x1
#define hipHostRegisterMapped 0x2
#define hipDeviceMallocDefault 0x0
#define hipDeviceMallocFinegrained 0x1
#define hipDeviceMallocUncached 0x3
#define hipIpcMemLazyEnablePeerAccess 0x1

extern "C" {
hipError_t hipGetDeviceCount(int* count);
hipError_t hipSetDevice(int device);
hipError_t hipGetDevice(int* device);
hipError_t hipDeviceSynchronize(void);
hipError_t hipDeviceGetPCIBusId(char* busid, int len, int device);
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned flags);
hipError_t hipDeviceCanAccessPeer(int* can, int device, int peer);
hipError_t hipMemGetInfo(size_t* free_bytes, size_t* total_bytes);
hipError_t hipGetLastError(void);
hipError_t hipPeekAtLastError(void);
const char* hipGetErrorString(hipError_t e);

hipError_t hipMalloc(void** ptr, size_t bytes);
hipError_t hipExtMallocWithFlags(void** ptr, size_t bytes, unsigned flags);
hipError_t hipFree(void* ptr);
hipError_t hipHostMalloc(void** ptr, size_t bytes, unsigned flags);
hipError_t hipHostFree(void* ptr);
hipError_t hipHostRegister(void* ptr, size_t bytes, unsigned flags);
hipError_t hipHostUnregister(void* ptr);
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned flags);
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* attr, const void* ptr);
hipError_t hipMemGetAddressRange(hipDeviceptr_t* base, size_t* bytes, hipDeviceptr_t ptr);

hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t stream);
hipError_t hipMemset(void* dst, int value, size_t bytes);
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t stream);

hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* handle, void* ptr);
hipError_t hipIpcOpenMemHandle(void** ptr, hipIpcMemHandle_t handle, unsigned flags);
hipError_t hipIpcCloseMemHandle(void* ptr);

hipError_t hipStreamCreate(hipStream_t* stream);
hipError_t hipStreamCreateWithFlags(hipStream_t* stream, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t stream);
hipError_t hipStreamSynchronize(hipStream_t stream);
hipError_t hipStreamQuery(hipStream_t stream);
hipError_t hipStreamWaitEvent(hipStream_t stream, hipEvent_t event, unsigned flags);
hipError_t hipStreamBeginCapture(hipStream_t stream, hipStreamCaptureMode mode);
hipError_t hipStreamEndCapture(hipStream_t stream, hipGraph_t* graph);
hipError_t hipStreamIsCapturing(hipStream_t stream, hipStreamCaptureStatus* status);

hipError_t hipEventCreate(hipEvent_t* event);
hipError_t hipEventCreateWithFlags(hipEvent_t* event, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t event);
hipError_t hipEventRecord(hipEvent_t event, hipStream_t stream);
hipError_t hipEventQuery(hipEvent_t event);
hipError_t hipEventSynchronize(hipEvent_t event);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t start, hipEvent_t stop);

hipError_t hipGraphInstantiate(hipGraphExec_t* exec, hipGraph_t graph, hipGraphNode_t* error_node, char* log, size_t log_bytes);
hipError_t hipGraphLaunch(hipGraphExec_t exec, hipStream_t stream);
hipError_t hipGraphDestroy(hipGraph_t graph);
hipError_t hipGraphExecDestroy(hipGraphExec_t exec);
}
