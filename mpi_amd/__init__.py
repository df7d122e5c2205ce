"""mpi_amd -- MI355X-native message passing and collectives behind btracey/mpi's API.

  csrc/        HIP kernels (gfx950) + engine + the C ABI of include/xmpi.h  -> libxmpi.so
  host/        C++ mirror of the reference's Go package (mpi.hpp): Interface, Register, flags,
               XGMI backend, Bcast/Reduce/Allreduce/Allgather                -> libxmpi_host.so
  xmpi.py      ctypes binding of the C ABI used by tests/ and bench.py (plumbing only)
  build.py     in-tree build of all of the above
"""
