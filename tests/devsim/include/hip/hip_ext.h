// tests/devsim (TEST INFRASTRUCTURE; see hip_runtime_api.h): a launch that carries its own begin / end events
#pragma once
#include "hip_runtime.h"

#define hipExtLaunchKernelGGL(kern, grid, block, shmem, stream, ev_start, ev_stop, flags, ...) \
  ::devsim::launch(kern, grid, block, stream, ev_start, ev_stop, #kern, __VA_ARGS__)
