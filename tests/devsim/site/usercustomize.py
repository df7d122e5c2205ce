"""tests/devsim (TEST INFRASTRUCTURE): with this directory on PYTHONPATH and XMPI_DEVSIM_LIB set, every Python process -- bench.py as
the driver launches it, the `--probe` children it starts, torch.distributed.run's workers -- binds mpi_amd.xmpi to the CPU stand-in
instead of libxmpi.so.  The product has no such switch: this file is the switch, and only the CPU suite puts it on the path."""
import os
import sys

if os.environ.get("XMPI_DEVSIM_LIB"):
    _root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    if _root not in sys.path:
        sys.path.insert(0, _root)
    from mpi_amd import xmpi as _xmpi
    _xmpi.LIB_PATH = os.environ["XMPI_DEVSIM_LIB"]
